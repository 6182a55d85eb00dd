// batch.cpp -- the lane behind ZK_LAUNCH (batch.h): members as fibers, their launches merged.
#include "batch.h"
#include "context.h"
#include "kernels.h"

#include <sys/mman.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace zk {

thread_local LaunchBatcher* t_batcher = nullptr;

namespace {
// pinned rings and fiber stacks cost milliseconds to make: finished batchers hand theirs back
std::mutex g_pool_mu;
std::vector<uint8_t*> g_rings;
std::vector<void*> g_stacks;
// one pinned allocation: [argument ring | staging up | staging down]
constexpr size_t ARG_BYTES = (size_t)8 << 20, UP_BYTES = (size_t)8 << 20, DOWN_BYTES = (size_t)48 << 20;
constexpr size_t RING_BYTES = ARG_BYTES + UP_BYTES + DOWN_BYTES;
// a fiber's stack: [guard page | 2 MiB | guard page] -- an overflow AND an underflow fault instead of corrupting a neighbour.  The
// provers keep their big arrays in std::vector; HIP runtime calls (hipMalloc in ctx_reserve, lazy plan builds) run on these stacks too,
// which is what the size is for (measured high-water mark of the recursion and transcript batches: zkhip_lockstep_stack_high_water).
constexpr size_t GUARD_BYTES = 4096, STACK_USABLE = (size_t)2 << 20, STACK_BYTES = STACK_USABLE + 2 * GUARD_BYTES;
std::atomic<uint64_t> g_stack_high_water{0};
// bytes of the usable area that were ever touched, from the resident pages (the area is fresh zero-fill mmap memory; a stack grows down)
size_t stack_touched(void* stack) {
    unsigned char vec[STACK_USABLE / 4096];
    if (mincore((char*)stack + GUARD_BYTES, STACK_USABLE, vec) != 0) return 0;
    size_t first = 0;
    while (first < sizeof(vec) && !(vec[first] & 1)) first++;
    return STACK_USABLE - first * 4096;
}
using Clock = std::chrono::steady_clock;
uint64_t ns_since(Clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count(); }
}  // namespace
std::atomic<uint64_t> g_lockstep_stats[6];                     // launches, requests, mixed rounds; ns: issuing launches, waiting for the stream, members' host code
uint64_t lockstep_stack_high_water() { return g_stack_high_water.load(); }

LaunchBatcher::LaunchBatcher(int members, hipStream_t stream) : members_(members), stream_(stream) {
    if (members < 1 || members > MAX_MEMBERS) return;
    req_.resize((size_t)members);
    fibers_.resize((size_t)members);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_rings.empty()) { ring_ = g_rings.back(); g_rings.pop_back(); }
        for (auto& f : fibers_) if (!g_stacks.empty()) { f.stack = g_stacks.back(); g_stacks.pop_back(); }
    }
    stacks_ok_ = true;
    for (auto& f : fibers_) {
        if (f.stack) continue;
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (p == MAP_FAILED) { stacks_ok_ = false; continue; }
        (void)mprotect(p, GUARD_BYTES, PROT_NONE);               // guard pages at BOTH ends
        (void)mprotect((char*)p + GUARD_BYTES + STACK_USABLE, GUARD_BYTES, PROT_NONE);
        f.stack = p;
    }
    if (ring_ || !stream_) return;                           // (no stream: the scheduler alone, see zkhip_selftest_lockstep)
    void* p = nullptr;
    unsigned flags = hipHostMallocPortable | hipHostMallocMapped;
#ifdef ZKHIP_AB_HOOKS
    static const int ring_nc = getenv("ZKHIP_RING_NC") ? atoi(getenv("ZKHIP_RING_NC")) : 0;
    if (ring_nc) flags |= hipHostMallocNonCoherent;
#endif
    if (hipHostMalloc(&p, RING_BYTES, flags) != hipSuccess) { (void)hipGetLastError(); return; }
    ring_ = (uint8_t*)p;
}

LaunchBatcher::~LaunchBatcher() {
    if (stream_) {
        g_lockstep_stats[0] += launches; g_lockstep_stats[1] += requests; g_lockstep_stats[2] += mixed;
        g_lockstep_stats[3] += flush_ns; g_lockstep_stats[4] += sync_ns; g_lockstep_stats[5] += host_ns;
    }
    // the last launches may still read the ring -- on the shared stream, or on a member's own stream ("loose" launches)
    if (ring_) { if (loose_used_) (void)hipDeviceSynchronize(); else (void)hipStreamSynchronize(stream_); }
    if (ran_) {
        uint64_t hw = 0;
        for (auto& f : fibers_) if (f.stack) { const uint64_t t = stack_touched(f.stack); if (t > hw) hw = t; }
        uint64_t cur = g_stack_high_water.load();
        while (hw > cur && !g_stack_high_water.compare_exchange_weak(cur, hw)) {}
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (ring_) { if (g_rings.size() < 16) g_rings.push_back(ring_); else (void)hipHostFree(ring_); }
    for (auto& f : fibers_)
        if (f.stack) { if (g_stacks.size() < 512) g_stacks.push_back(f.stack); else (void)munmap(f.stack, STACK_BYTES); }
}

// ---- the lane: runs the members until each has returned; when none can run, somebody's wait is over
void LaunchBatcher::trampoline() {
    LaunchBatcher* self = t_batcher;
    const int b = self->current_;
    // nothing may unwind past this frame (there is no caller above it: uc_link is null): an exception that escapes a member --
    // std::bad_alloc from a prover's vectors, std::system_error from a mutex -- is caught by the member function the dealer hands in (jobs.cpp:
    // that member's status, with its text); what still arrives here (a member function that does not catch) marks the whole batch, not the process
    try { (*self->fn_)(b); }
    catch (const std::exception& e) { self->sticky_ = hipErrorUnknown; set_error(std::string("lock-step member: ") + e.what()); }
    catch (...) { self->sticky_ = hipErrorUnknown; set_error("lock-step member: unknown exception"); }
    self->fibers_[(size_t)b].state = DONE;
    swapcontext(&self->fibers_[(size_t)b].ctx, &self->lane_);  // never resumed
}

void LaunchBatcher::switch_to(int b) {
    Fiber& f = fibers_[(size_t)b];
    current_ = b;
    fiber_tls_swap_error(f.tls.last_error);
    const auto t0 = Clock::now();
    swapcontext(&lane_, &f.ctx);
    host_ns += ns_since(t0);
    fiber_tls_swap_error(f.tls.last_error);
    current_ = -1;
}

void LaunchBatcher::park(State s) {
    Fiber& f = fibers_[(size_t)current_];
    f.state = s;
    swapcontext(&f.ctx, &lane_);
}

void LaunchBatcher::run(const std::function<void(int)>& fn) {
    fn_ = &fn;
    ran_ = true;
    LaunchBatcher* outer = t_batcher;
    t_batcher = this;
    for (int b = 0; b < members_; b++) {
        Fiber& f = fibers_[(size_t)b];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = (char*)f.stack + GUARD_BYTES;
        f.ctx.uc_stack.ss_size = STACK_USABLE;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())&LaunchBatcher::trampoline, 0);
        f.state = RUNNABLE;
    }
    for (;;) {
        bool ran = false, live = false;
        for (int b = 0; b < members_; b++) {
            if (fibers_[(size_t)b].state == RUNNABLE) { switch_to(b); ran = true; }
            if (fibers_[(size_t)b].state != DONE) live = true;
        }
        if (!live) break;
        if (!ran) resolve();
    }
    t_batcher = outer;
    fn_ = nullptr;
}

// Nobody can run.  Pending launches go first; else the members that want the stream drained get that; else everybody still in the
// batch is voting.  Members whose sequences differ for a while (a fresh context sets up its key while the others already prove)
// therefore never wait for each other in a cycle.
void LaunchBatcher::resolve() {
    int idx[MAX_MEMBERS], n = 0;
    for (int b = 0; b < members_; b++) if (fibers_[(size_t)b].state == AT_LAUNCH) idx[n++] = b;
    if (n > 0) {
        const auto t0 = Clock::now();
        bool taken[MAX_MEMBERS] = {};
        int groups = 0;
        for (int i = 0; i < n; i++) {
            if (taken[i]) continue;
            const Request& a = req_[(size_t)idx[i]];
            int grp[MAX_MEMBERS], cnt = 0;
            for (int j = i; j < n; j++) {
                const Request& r = req_[(size_t)idx[j]];
                if (!taken[j] && r.fn == a.fn && r.size == a.size && r.lds == a.lds && r.grid.x == a.grid.x && r.grid.y == a.grid.y &&
                    r.grid.z == a.grid.z && r.block.x == a.block.x && r.block.y == a.block.y && r.block.z == a.block.z) { taken[j] = true; grp[cnt++] = idx[j]; }
            }
            groups++;
            const hipError_t e = launch_group(a, grp, cnt, nullptr, stream_);
            if (e != hipSuccess) sticky_ = e;
            for (int k = 0; k < cnt; k++) { fibers_[(size_t)grp[k]].status = e; fibers_[(size_t)grp[k]].state = RUNNABLE; }
        }
        if (groups > 1) mixed++;
        flush_ns += ns_since(t0);
        return;
    }
    for (int b = 0; b < members_; b++) if (fibers_[(size_t)b].state == AT_SYNC) idx[n++] = b;
    if (n > 0) {
        const auto t0 = Clock::now();
        const hipError_t e = stream_ ? hipStreamSynchronize(stream_) : hipSuccess;
        if (e != hipSuccess) sticky_ = e;
        for (int k = 0; k < n; k++) { fibers_[(size_t)idx[k]].status = e; fibers_[(size_t)idx[k]].state = RUNNABLE; }
        sync_ns += ns_since(t0);
        return;
    }
    for (int b = 0; b < members_; b++) if (fibers_[(size_t)b].state == AT_HOST) idx[n++] = b;
    if (n > 0) {
        // the members' host work of one kind, side by side (groups of equal (fn, n); a member that came alone is served alone)
        bool taken[MAX_MEMBERS] = {};
        for (int i = 0; i < n; i++) {
            if (taken[i]) continue;
            const HostOp& a = fibers_[(size_t)idx[i]].host;
            void* objs[MAX_MEMBERS]; const uint32_t* words[MAX_MEMBERS];
            int cnt = 0;
            for (int j = i; j < n; j++) {
                const HostOp& r = fibers_[(size_t)idx[j]].host;
                if (!taken[j] && r.fn == a.fn && r.n == a.n) { taken[j] = true; objs[cnt] = r.obj; words[cnt] = r.words; cnt++; }
            }
            a.fn(objs, words, a.n, cnt);
            host_merges++;
        }
        for (int k = 0; k < n; k++) fibers_[(size_t)idx[k]].state = RUNNABLE;
        return;
    }
    bool verdict = true;
    for (int b = 0; b < members_; b++) if (fibers_[(size_t)b].state == AT_VOTE) { idx[n++] = b; verdict = verdict && fibers_[(size_t)b].vote; }
    vote_result_ = verdict;
    for (int k = 0; k < n; k++) fibers_[(size_t)idx[k]].state = RUNNABLE;
    if (n == 0) {                                               // cannot happen: a live fiber is runnable or parked in one of the four states
        sticky_ = hipErrorUnknown;
        for (auto& f : fibers_) if (f.state != DONE) f.state = RUNNABLE;
    }
}

// cnt requests of one kind (members idx, or one loose request): their arguments into the ring, one launch with gridDim.z = cnt
hipError_t LaunchBatcher::launch_group(const Request& a, const int* idx, int cnt, const void* loose, hipStream_t s) {
    if (a.grid.z != 1) return hipErrorInvalidValue;             // z is the batch index
    const size_t stride = a.size, bytes = (stride * (size_t)cnt + 255) & ~(size_t)255;
    hipError_t status = hipSuccess;
    if (ring_pos_ + bytes > ARG_BYTES) {                         // the ring wraps: earlier launches may still read it
        status = loose_used_ ? hipDeviceSynchronize() : hipStreamSynchronize(stream_);
        ring_pos_ = 0;
    }
    uint8_t* slot = ring_ + ring_pos_;
    ring_pos_ += bytes;
    if (s != stream_) loose_used_ = true;
    if (loose) std::memcpy(slot, loose, stride);
    else for (int k = 0; k < cnt; k++) std::memcpy(slot + stride * (size_t)k, req_[(size_t)idx[k]].args, stride);
    void* karg = slot;
    void* kargs[1] = {&karg};
    const hipError_t e = hipLaunchKernel(a.fn, dim3(a.grid.x, a.grid.y, (unsigned)cnt), a.block, kargs, a.lds, s);
    launches++;
    return e != hipSuccess ? e : status;
}

hipError_t LaunchBatcher::launch(const void* batch_kernel, const void* args, size_t size, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    requests++;
    if (!ring_ || size > MAX_ARGS || current_ < 0) { sticky_ = hipErrorInvalidValue; return sticky_; }
    Request& r = req_[(size_t)current_];
    r.fn = batch_kernel; r.grid = grid; r.block = block; r.lds = lds; r.size = size;
    if (s != stream_) {                                          // not on the shared stream: nothing to merge with, launched at once
        const hipError_t e = launch_group(r, nullptr, 1, args, s);
        if (e != hipSuccess) sticky_ = e;
        return e;
    }
    std::memcpy(r.args, args, size);
    const int me = current_;
    park(AT_LAUNCH);
    return fibers_[(size_t)me].status;
}

bool LaunchBatcher::all(bool mine) {
    if (current_ < 0) return mine;
    fibers_[(size_t)current_].vote = mine;
    park(AT_VOTE);
    return vote_result_;
}

void LaunchBatcher::host_merge(HostMergeFn fn, void* obj, const uint32_t* words, size_t n) {
    if (current_ < 0) { fn(&obj, &words, n, 1); return; }
#ifdef ZKHIP_AB_HOOKS
    static const bool serial = getenv("ZKHIP_HOST_MERGE") && atoi(getenv("ZKHIP_HOST_MERGE")) == 0;      // A/B: every member on its own, as before round 6
    if (serial) { fn(&obj, &words, n, 1); return; }
#endif
    HostOp& h = fibers_[(size_t)current_].host;
    h.fn = fn; h.obj = obj; h.words = words; h.n = n;
    park(AT_HOST);
}

hipError_t LaunchBatcher::sync_all() {
    if (current_ < 0) return stream_ ? hipStreamSynchronize(stream_) : hipSuccess;
    const int me = current_;
    park(AT_SYNC);
    return fibers_[(size_t)me].status;
}

// (A hint only: a member that has left never parks again, so the lane runs it to its end like any runnable fiber and the votes of
// the others -- taken when nobody can run -- never wait for it.)
void LaunchBatcher::leave() {}

// A member's staging areas are rings of its own inside the pinned block.  up: the data is consumed by a kernel launched after the
// member wrote it, so a wrap waits for the stream; down: written by a kernel launched after the member read what it was given
// earlier, so a wrap needs no wait.  A copy of more than half a region is the caller's to do on its own.
uint8_t* LaunchBatcher::stage(size_t bytes, size_t base, size_t region, size_t& pos, bool sync_on_wrap) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!ring_ || current_ < 0 || need > region / 2) return nullptr;
    if (pos + need > region) {
        if (sync_on_wrap) { const hipError_t e = hipStreamSynchronize(stream_); if (e != hipSuccess) sticky_ = e; }
        pos = 0;
    }
    uint8_t* p = ring_ + base + (size_t)current_ * region + pos;
    pos += need;
    return p;
}
uint8_t* LaunchBatcher::stage_up(size_t bytes) {
    const size_t region = (UP_BYTES / (size_t)members_) & ~(size_t)63;
    return current_ < 0 ? nullptr : stage(bytes, ARG_BYTES, region, fibers_[(size_t)current_].up_pos, true);
}
uint8_t* LaunchBatcher::stage_down(size_t bytes) {
    const size_t region = (DOWN_BYTES / (size_t)members_) & ~(size_t)63;
    return current_ < 0 ? nullptr : stage(bytes, ARG_BYTES + UP_BYTES, region, fibers_[(size_t)current_].down_pos, false);
}

// The lane's scheduler on its own (no device, no stream): `members` fibers run `rounds` rounds of {set the thread's error string to
// a value of their own, wait for the stream, vote, check that the string is still theirs}; one member votes "no" in one round and
// every member must see that round's verdict; members leave at different rounds.  0, or the number of the first check that failed.
int lockstep_selftest(int members, int rounds) {
    if (members < 1 || members > LaunchBatcher::MAX_MEMBERS || rounds < 1) return 1;
    LaunchBatcher lb(members, nullptr);
    if (!lb.ok_scheduler()) return 2;
    std::vector<int> bad((size_t)members, 0), done((size_t)members, 0);
    const int no_round = rounds / 2, no_member = members / 3;
    lb.run([&](int b) {
        const int mine = rounds - (b % 3 == 2 ? rounds / 4 : 0);          // a third of the members leave early
        for (int r = 0; r < mine; r++) {
            const std::string tag = "member " + std::to_string(b) + " round " + std::to_string(r);
            set_error(tag);
            if (lb.sync_all() != hipSuccess) bad[(size_t)b] = 3;
            if (tag != zkhip_last_error()) bad[(size_t)b] = 4;             // another fiber's string leaked into this one
            if (r < rounds - rounds / 4) {                                 // rounds in which every member is still here: a vote
                const bool verdict = lb.all(!(r == no_round && b == no_member));
                if (verdict != (r != no_round)) bad[(size_t)b] = 5;
                if (tag != zkhip_last_error()) bad[(size_t)b] = 6;
            }
        }
        done[(size_t)b] = 1;
    });
    for (int b = 0; b < members; b++) { if (bad[(size_t)b]) return bad[(size_t)b]; if (!done[(size_t)b]) return 7; }
    return 0;
}

// ---- the prover's copies, memsets and waits: plain stream operations, or their merged forms inside a lock-step batch
uint32_t coop_max_nodes() {
    if (!t_batcher) return COOP_MAX_NODES;
#ifdef ZKHIP_AB_HOOKS
    static const int keep = getenv("ZKHIP_COOP_KEEP") ? atoi(getenv("ZKHIP_COOP_KEEP")) : 0;
    if (keep) return COOP_MAX_NODES;
#endif
    const uint32_t m = (uint32_t)t_batcher->members(), v = COOP_MAX_NODES / (m ? m : 1u);      // (a bound 2x / 4x lower: 1 - 2 ms slower)
    return v < COOP_TOP_NODES ? COOP_TOP_NODES : v;
}
int dev_sync(zkhip_ctx* ctx) {
    const hipError_t e = t_batcher && t_batcher->stream() == ctx->stream ? t_batcher->sync_all() : hipStreamSynchronize(ctx->stream);
    return e == hipSuccess ? ZKHIP_OK : hip_fail(e, "stream synchronize");
}
int dev_d2h(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ZKHIP_OK;
    uint8_t* st = t_batcher && t_batcher->stream() == ctx->stream ? t_batcher->stage_down(bytes) : nullptr;
    hipError_t e;
    if (st) {
        e = launch_copy_bytes(st, src, bytes, ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "copy to host (staged)");
        const int rc = dev_sync(ctx);
        if (rc != ZKHIP_OK) return rc;
        std::memcpy(dst, st, bytes);
        return ZKHIP_OK;
    }
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync (to host)");
    return dev_sync(ctx);
}
int dev_h2d(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ZKHIP_OK;
    uint8_t* st = t_batcher && t_batcher->stream() == ctx->stream ? t_batcher->stage_up(bytes) : nullptr;
    hipError_t e;
    if (st) {                                                   // the staged copy is the member's own: no wait for the stream
        std::memcpy(st, src, bytes);
        e = launch_copy_bytes(dst, st, bytes, ctx->stream);
        return e == hipSuccess ? ZKHIP_OK : hip_fail(e, "copy to device (staged)");
    }
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync (to device)");
    return dev_sync(ctx);                                       // the source is pageable host memory that may go out of scope
}
int dev_memset(zkhip_ctx* ctx, void* dst, int byte, size_t bytes) {
    if (bytes == 0) return ZKHIP_OK;
    const hipError_t e = t_batcher && t_batcher->stream() == ctx->stream ? launch_fill_bytes(dst, byte, bytes, ctx->stream)
                                                                         : hipMemsetAsync(dst, byte, bytes, ctx->stream);
    return e == hipSuccess ? ZKHIP_OK : hip_fail(e, "memset");
}

}  // namespace zk
