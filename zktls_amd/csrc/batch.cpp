// batch.cpp -- the rendezvous behind ZK_LAUNCH (batch.h).
#include "batch.h"
#include "context.h"
#include "kernels.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace zk {

thread_local LaunchBatcher* t_batcher = nullptr;

// argument rings are pinned host memory: allocating one costs milliseconds, so finished batchers hand theirs back
namespace {
std::mutex g_ring_mu;
std::vector<uint8_t*> g_rings;
// one pinned allocation: [argument ring | staging up | staging down]
constexpr size_t ARG_BYTES = (size_t)8 << 20, UP_BYTES = (size_t)8 << 20, DOWN_BYTES = (size_t)48 << 20;
constexpr size_t RING_BYTES = ARG_BYTES + UP_BYTES + DOWN_BYTES;
}  // namespace
std::atomic<uint64_t> g_lockstep_stats[6];      // launches, requests, mixed; ns: waiting at a rendezvous, flushing, in votes

LaunchBatcher::LaunchBatcher(int members, hipStream_t stream) : members_(members), stream_(stream) {
    if (members < 1 || members > MAX_MEMBERS) return;
    req_.resize((size_t)members);
    spin_ = std::thread::hardware_concurrency() >= 2u * (unsigned)members + 8u;
    if (const char* e = getenv("ZKHIP_LOCKSTEP_SPIN")) spin_ = atoi(e) != 0;
    ring_size_ = ARG_BYTES;
    {
        std::lock_guard<std::mutex> lk(g_ring_mu);
        if (!g_rings.empty()) { ring_ = g_rings.back(); g_rings.pop_back(); }
    }
    if (ring_) return;
    void* p = nullptr;
    if (hipHostMalloc(&p, RING_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
    ring_ = (uint8_t*)p;
}

LaunchBatcher::~LaunchBatcher() {
    g_lockstep_stats[0] += launches; g_lockstep_stats[1] += requests; g_lockstep_stats[2] += mixed;
    g_lockstep_stats[3] += wait_ns; g_lockstep_stats[4] += flush_ns; g_lockstep_stats[5] += vote_ns;
    if (!ring_) return;
    (void)hipStreamSynchronize(stream_);                       // the last launches may still read the ring
    std::lock_guard<std::mutex> lk(g_ring_mu);
    if (g_rings.size() < 16) g_rings.push_back(ring_);
    else (void)hipHostFree(ring_);
}

// cnt requests of one kind (idx into req_, or one loose request): their arguments into the ring, one launch with gridDim.z = cnt
hipError_t LaunchBatcher::launch_group(const Request& a, const int* idx, int cnt, const void* loose, hipStream_t s) {
    if (a.grid.z != 1) return hipErrorInvalidValue;             // z is the batch index
    const size_t stride = a.size, bytes = (stride * (size_t)cnt + 255) & ~(size_t)255;
    hipError_t status = hipSuccess;
    if (ring_pos_ + bytes > ring_size_) {                        // the ring wraps: earlier launches may still read it
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

// all members that will still launch are here (mu_ held): one launch per group of identical requests, then release them
hipError_t LaunchBatcher::flush_locked() {
    hipError_t status = hipSuccess;
    const int n = arrived_;
    bool taken[MAX_MEMBERS] = {};
    int groups = 0;
    for (int i = 0; i < n; i++) {
        if (taken[i]) continue;
        const Request& a = req_[(size_t)i];
        int idx[MAX_MEMBERS], cnt = 0;
        for (int j = i; j < n; j++) {
            const Request& b = req_[(size_t)j];
            if (!taken[j] && b.fn == a.fn && b.size == a.size && b.lds == a.lds && b.grid.x == a.grid.x && b.grid.y == a.grid.y &&
                b.grid.z == a.grid.z && b.block.x == a.block.x && b.block.y == a.block.y && b.block.z == a.block.z) { taken[j] = true; idx[cnt++] = j; }
        }
        groups++;
        const hipError_t e = launch_group(a, idx, cnt, nullptr, stream_);
        if (e != hipSuccess) status = e;
    }
    if (groups > 1) mixed++;
    if (getenv("ZKHIP_LOCKSTEP_DEBUG")) {
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[launch %llu] at %.3f ms, first arrival %.3f ms earlier, grid %u block %u\n", (unsigned long long)launches.load(),
            std::chrono::duration<double, std::milli>(now - born_).count(), std::chrono::duration<double, std::milli>(now - first_arrival_).count(), req_[0].grid.x, req_[0].block.x);
        if (std::chrono::duration<double, std::milli>(now - first_arrival_).count() > 2.0) {
            std::fprintf(stderr, "   arrivals (ms):");
            for (int i = 0; i < n; i++) std::fprintf(stderr, " %.1f", dbg_arrival_[i]);
            std::fprintf(stderr, "\n");
        }
    }
    arrived_ = 0;
    gen_status_ = status;
    if (status != hipSuccess) sticky_ = status;
    gen_++;
    cv_.notify_all();
    return status;
}

hipError_t LaunchBatcher::launch(const void* batch_kernel, const void* args, size_t size, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    std::unique_lock<std::mutex> lk(mu_);
    requests++;
    if (!ring_ || size > MAX_ARGS) { sticky_ = hipErrorInvalidValue; return sticky_; }
    if (s != stream_) {                                          // not on the shared stream: nothing to merge with, launched at once
        Request r;
        r.fn = batch_kernel; r.grid = grid; r.block = block; r.lds = lds; r.size = size;
        const hipError_t e = launch_group(r, nullptr, 1, args, s);
        if (e != hipSuccess) sticky_ = e;
        return e;
    }
    if (arrived_ == 0) first_arrival_ = std::chrono::steady_clock::now();
    dbg_arrival_[arrived_] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - first_arrival_).count();
    Request& r = req_[(size_t)arrived_++];
    r.fn = batch_kernel; r.grid = grid; r.block = block; r.lds = lds; r.size = size;
    std::memcpy(r.args, args, size);
    const auto t0 = std::chrono::steady_clock::now();
    if (arrived_ + votes_ + syncers_ >= members_) {
        const hipError_t e = flush_locked();
        flush_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        return e;
    }
    const uint64_t g = gen_.load(std::memory_order_relaxed);
    wait_for(lk, gen_, g);
    wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return gen_status_;
}
// mu_ held on entry, released on return.  With a hardware thread per member the waiters spin on the counter (woken within a
// fraction of a microsecond and without touching the mutex: sixty-four sleepers re-taking one mutex cost ~0.5 ms per rendezvous);
// otherwise, or when the others take long (host work between launches), they sleep on the condition variable.
void LaunchBatcher::wait_for(std::unique_lock<std::mutex>& lk, const std::atomic<uint64_t>& counter, uint64_t seen) {
    if (spin_) {
        lk.unlock();
        for (int i = 0; i < 200000; i++) {
            if (counter.load(std::memory_order_acquire) != seen) return;
            __builtin_ia32_pause();
        }
        lk.lock();
    }
    while (!cv_.wait_for(lk, std::chrono::seconds(20), [&] { return counter.load(std::memory_order_acquire) != seen; })) {
        // nobody moved for twenty seconds: some member is stuck outside the batcher (or the launch sequences diverged for good)
        std::fprintf(stderr, "[zkhip lock-step] still waiting: %d members, %d at a launch, %d syncing, %d voting\n", members_, arrived_, syncers_, votes_);
    }
    lk.unlock();
}

// When every member waits somewhere (mu_ held) somebody has to move: pending launches go first; else the members that want the
// stream drained get that (harmless whoever else is elsewhere); a vote needs every member.  Members whose sequences differ for a
// while (a fresh context sets up its key while the others already prove) therefore never wait for each other in a cycle.
void LaunchBatcher::resolve_locked() {
    if (members_ <= 0) return;
    if (arrived_ > 0 && arrived_ + votes_ + syncers_ >= members_) (void)flush_locked();
    if (syncers_ > 0 && arrived_ == 0 && syncers_ + votes_ >= members_) {
        const auto s0 = std::chrono::steady_clock::now();
        const hipError_t e = hipStreamSynchronize(stream_);
        const double sms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s0).count();
        if (getenv("ZKHIP_LOCKSTEP_DEBUG")) std::fprintf(stderr, "[sync] after %llu launches: %.3f ms, since batch start %.3f ms\n", (unsigned long long)launches.load(), sms,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - born_).count());
        if (e != hipSuccess) sticky_ = e;
        sync_status_ = e; syncers_ = 0; sync_gen_++;
        cv_.notify_all();
    }
    if (votes_ > 0 && votes_ >= members_) {
        vote_result_ = vote_and_; vote_and_ = true; votes_ = 0; vote_gen_++;
        cv_.notify_all();
    }
}

bool LaunchBatcher::all(bool mine) {
    std::unique_lock<std::mutex> lk(mu_);
    vote_and_ = vote_and_ && mine;
    votes_++;
    const uint64_t g = vote_gen_.load(std::memory_order_relaxed);
    resolve_locked();
    if (vote_gen_.load(std::memory_order_relaxed) != g) return vote_result_;
    wait_for(lk, vote_gen_, g);
    return vote_result_;
}

hipError_t LaunchBatcher::sync_all() {
    std::unique_lock<std::mutex> lk(mu_);
    const auto t0 = std::chrono::steady_clock::now();
    syncers_++;
    const uint64_t g = sync_gen_.load(std::memory_order_relaxed);
    resolve_locked();
    if (sync_gen_.load(std::memory_order_relaxed) == g) wait_for(lk, sync_gen_, g);
    vote_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return sync_status_;
}

void LaunchBatcher::leave() {
    std::unique_lock<std::mutex> lk(mu_);
    members_--;
    resolve_locked();
}

// A staging area is a ring of its own.  up: the data is consumed by a kernel launched after the member wrote it, so a wrap waits for
// the stream; down: written by a kernel that cannot be launched before every member has copied out what it was given earlier (all of
// them must arrive at that launch), so a wrap needs no wait -- as long as one generation fits (callers: a quarter of the ring each)
uint8_t* LaunchBatcher::stage(size_t bytes, size_t base, size_t size, size_t& pos, bool sync_on_wrap) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!ring_ || need * (size_t)(members_ > 0 ? members_ : 1) > size / 2) return nullptr;
    std::unique_lock<std::mutex> lk(mu_);
    if (pos + need > size) {
        if (sync_on_wrap) { const hipError_t e = hipStreamSynchronize(stream_); if (e != hipSuccess) sticky_ = e; }
        pos = 0;
    }
    uint8_t* p = ring_ + base + pos;
    pos += need;
    return p;
}
uint8_t* LaunchBatcher::stage_up(size_t bytes) { return stage(bytes, ARG_BYTES, UP_BYTES, up_pos_, true); }
uint8_t* LaunchBatcher::stage_down(size_t bytes) { return stage(bytes, ARG_BYTES + UP_BYTES, DOWN_BYTES, down_pos_, false); }

// ---- the prover's copies, memsets and waits: plain stream operations, or their merged forms inside a lock-step batch
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
