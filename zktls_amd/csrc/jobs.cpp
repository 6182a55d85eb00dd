// jobs.cpp -- the batch entries: many independent proofs in ONE call, dealt over the GPUs of the node (shard-parallel, no
// exchange step) on pooled contexts -- one context + stream + host thread per worker, or, for small jobs, lock-step lanes (batch.h).
// The reference proves its shards / transcripts inside one `client.prove` call (sp1.rs:116, prover.rs:90).
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "context.h"
#include "batch.h"

using namespace zk;

extern "C" {

// ---- a batch of independent shards, `in_flight` at a time (one internal context + host thread each)
// Contexts (HIP stream + multi-GiB workspaces) are expensive to create and to free (hipFree synchronises the device), so the
// batch entry keeps the ones it made in a process-wide pool per device; zkhip_release_cached_contexts() empties it.
namespace {
std::mutex g_pool_mu;
std::vector<std::pair<int, zkhip_ctx*>> g_pool;
zkhip_ctx* pool_take(int device) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
#ifdef ZKHIP_AB_HOOKS
    static const bool fifo = getenv("ZKHIP_POOL_FIFO") != nullptr;
    if (fifo)
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].first == device) { zkhip_ctx* c = g_pool[i].second; g_pool.erase(g_pool.begin() + (long)i); return c; }
#endif
    for (size_t i = g_pool.size(); i-- > 0;)                    // the context returned last first: its workspaces and keys fit the work that is running now
        if (g_pool[i].first == device) { zkhip_ctx* c = g_pool[i].second; g_pool.erase(g_pool.begin() + (long)i); return c; }
    return nullptr;
}
void pool_give(int device, zkhip_ctx* c) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.emplace_back(device, c);
}
size_t ctx_workspace_bytes(const zkhip_ctx* c) {
    size_t b = 0;
    for (int s = 0; s < S_COUNT; s++) b += c->scratch[s].bytes;
    return b;
}
size_t pool_bytes(int device) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t b = 0;
    for (auto& e : g_pool) if (e.first == device) b += ctx_workspace_bytes(e.second);
    return b;
}
// A lock-step call may leave lanes x members contexts behind.  The pool keeps the ones returned last while their workspaces stay within
// `budget` bytes and their number within `max_count`; the others are destroyed (outside the lock: hipFree synchronises the device).
void pool_trim(int device, size_t budget, size_t max_count) {
    std::vector<zkhip_ctx*> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t kept = 0, bytes = 0;
        for (size_t i = g_pool.size(); i-- > 0;) {
            if (g_pool[i].first != device) continue;
            const size_t b = ctx_workspace_bytes(g_pool[i].second);
            if (kept + 1 > max_count || bytes + b > budget) { drop.push_back(g_pool[i].second); g_pool.erase(g_pool.begin() + (long)i); }
            else { kept++; bytes += b; }
        }
    }
    for (zkhip_ctx* c : drop) zkhip_ctx_destroy(c);
}
}  // namespace
void zkhip_release_cached_contexts(void) {
    std::vector<std::pair<int, zkhip_ctx*>> all;
    { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
    for (auto& e : all) zkhip_ctx_destroy(e.second);
    zk::rec_release_host_tables();
}
// Job i of a batch goes to devices[i mod n_devices] (shard-parallel, no exchange step); every device runs up to
// `in_flight` workers (a pooled context + HIP stream + host thread each) that take that device's jobs in index order and call
// run(ctx, i) -> status.  Returns the status of the lowest failing job (its message in zkhip_last_error), and tells through `ran`
// which jobs a worker reached at all (none when every worker of a device failed to get a context).
extern "C++" {
namespace zk {
int deal_jobs(const int* devices, int n_devices, int n_jobs, int in_flight, const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran) {
    ran.assign((size_t)n_jobs, 0);
    if (n_jobs == 0) return ZKHIP_OK;
    if (in_flight <= 0) in_flight = 4;
    std::vector<std::atomic<int>> next(n_devices);            // per device: how many of ITS jobs were handed out
    for (auto& a : next) a.store(0);
    std::mutex mu;
    int first_rc = ZKHIP_OK, first_job = n_jobs, ctx_rc = ZKHIP_OK;
    std::string first_msg, ctx_msg;
    auto note = [&](int job, int rc) {                     // keep the failure of the lowest job index
        std::lock_guard<std::mutex> lk(mu);
        if (job < first_job) { first_job = job; first_rc = rc; first_msg = zkhip_last_error(); }
    };
    auto worker = [&](int slot) {
        const int device = devices[slot];
        zkhip_ctx* ctx = pool_take(device);
        int rc = ctx ? ZKHIP_OK : zkhip_ctx_create(device, nullptr, &ctx);
        if (rc != ZKHIP_OK) {                                // e.g. no memory for one more workspace: the other workers carry on
            std::lock_guard<std::mutex> lk(mu);
            ctx_rc = rc; ctx_msg = zkhip_last_error();
            return;
        }
        bool healthy = true;                                 // a context that saw a failing job does not go back to the pool:
        for (;;) {                                           // a sticky HIP error or a half-built key would fail unrelated jobs later
            const int k = next[slot].fetch_add(1);
            const long i = (long)slot + (long)k * n_devices;   // the k-th job of this device
            if (i >= n_jobs) break;
            rc = run(ctx, (int)i);
            ran[(size_t)i] = 1;
            if (rc != ZKHIP_OK) { note((int)i, rc); healthy = false; }
        }
        if (zkhip_ctx_sync(ctx) != ZKHIP_OK) healthy = false;
        if (healthy) pool_give(device, ctx);
        else zkhip_ctx_destroy(ctx);
    };
    std::vector<std::thread> pool;
    for (int slot = 0; slot < n_devices; slot++) {
        const int mine = (n_jobs - slot + n_devices - 1) / n_devices;       // jobs of this device
        const int workers = mine < in_flight ? mine : in_flight;
        for (int t = 0; t < workers; t++) {
            // thread creation can throw (resource limits): never let that unwind through joinable threads into the C ABI --
            // the workers already started (or, with none, this thread) take the jobs instead
            try { pool.emplace_back(worker, slot); }
            catch (...) { if (t == 0) worker(slot); break; }
        }
    }
    for (auto& t : pool) t.join();
    if (first_rc != ZKHIP_OK) { set_error(first_msg); return first_rc; }
    for (int i = 0; i < n_jobs; i++)                         // jobs nobody could take: every worker of that device failed to get a context
        if (!ran[(size_t)i]) { set_error(ctx_msg.empty() ? "prove_shards: job not run" : ctx_msg); return ctx_rc != ZKHIP_OK ? ctx_rc : ZKHIP_ERR_INVALID; }
    return ZKHIP_OK;
}
// lock-step batches (batch.h): members per batch (0 / 1: off) and batches in flight per device; process-wide
static std::atomic<int> g_lockstep_batch{16}, g_lockstep_lanes{6};
int lockstep_batch() { return g_lockstep_batch.load(); }
// the largest member (trace cells) that still goes through the lock-step lanes
uint64_t lockstep_max_cells() {
#ifdef ZKHIP_AB_HOOKS
    if (const char* e = getenv("ZKHIP_LOCKSTEP_MAX_CELLS")) return strtoull(e, nullptr, 0);
#endif
    return LOCKSTEP_MAX_CELLS;
}
int lockstep_lanes() { return g_lockstep_lanes.load(); }
void lockstep_set(int max_batch, int lanes) {
    g_lockstep_batch.store(max_batch < 0 ? 0 : (max_batch > LaunchBatcher::MAX_MEMBERS ? LaunchBatcher::MAX_MEMBERS : max_batch));
    if (lanes > 0) g_lockstep_lanes.store(lanes > 32 ? 32 : lanes);
}
struct HostPool::Impl {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> queue;
    std::vector<std::thread> threads;
    int running = 0;
    bool stop = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !queue.empty(); });
            if (queue.empty()) return;
            std::function<void()> job = std::move(queue.front());
            queue.pop_front();
            running++;
            lk.unlock();
            job();
            lk.lock();
            running--;
            if (queue.empty() && running == 0) idle.notify_all();
        }
    }
};
HostPool::HostPool(int threads) : impl_(new Impl) {
    for (int t = 0; t < threads; t++) {
        try { impl_->threads.emplace_back([this] { impl_->loop(); }); } catch (...) { break; }
    }
}
HostPool::~HostPool() {
    wait();
    { std::lock_guard<std::mutex> lk(impl_->mu); impl_->stop = true; }
    impl_->cv.notify_all();
    for (auto& t : impl_->threads) t.join();
    delete impl_;
}
void HostPool::submit(std::function<void()> job) {
    if (impl_->threads.empty()) { job(); return; }              // no thread could be made: the caller does it
    { std::lock_guard<std::mutex> lk(impl_->mu); impl_->queue.push_back(std::move(job)); }
    impl_->cv.notify_one();
}
void HostPool::wait() {
    std::unique_lock<std::mutex> lk(impl_->mu);
    impl_->idle.wait(lk, [&] { return impl_->queue.empty() && impl_->running == 0; });
}

// Lock-step variant for SMALL proofs (batch.h): the jobs of a device are grouped by `shape[i]` (jobs of one shape run the same launch
// sequence), groups are cut into batches of up to `max_batch`, and a batch is proven by that many provers on pooled contexts that
// share ONE stream and merge their kernel launches through a LaunchBatcher.  `lanes` batches are in flight per device (a host thread
// each; the members of a batch are fibers of that thread), so that the host-side work around one batch (padding, transcripts,
// verification) overlaps the other batches' kernels.  Same contract as deal_jobs.
int deal_jobs_lockstep(const int* devices, int n_devices, int n_jobs, const int* shape, int max_batch, int lanes,
                       const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran, uint64_t job_cells) {
    ran.assign((size_t)n_jobs, 0);
    if (n_jobs == 0) return ZKHIP_OK;
    if (max_batch < 1) max_batch = 1;
    if (max_batch > LaunchBatcher::MAX_MEMBERS) max_batch = LaunchBatcher::MAX_MEMBERS;
    if (lanes < 1) lanes = 1;
    // Memory budget.  Every member of every lane owns a context whose workspaces grow to what its proof needs (LOCKSTEP_BYTES_PER_CELL
    // per trace cell, the measured footprint of the provers at blowup 2 .. 4 with room to spare).  lanes x max_batch members must fit
    // in what the device has free plus what this process's pool already holds for it; otherwise the batches get smaller, then the
    // lanes fewer.  (A member that still runs out of memory is retried on its own below.)
    size_t pool_budget = (size_t)4 << 30;
    if (job_cells) {
        const size_t est = (size_t)job_cells * LOCKSTEP_BYTES_PER_CELL + ((size_t)16 << 20);
        int caller_dev = -1;                                       // the caller's current device is the caller's: put it back after the queries
        if (hipGetDevice(&caller_dev) != hipSuccess) { (void)hipGetLastError(); caller_dev = -1; }
        for (int d = 0; d < n_devices; d++) {
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(physical_device(devices[d])) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); continue; }
            const size_t allow = free_b / 10 * 8 + pool_bytes(devices[d]);
            size_t cap = allow / est;
            if (cap < 2) cap = 2;
            while ((size_t)lanes * (size_t)max_batch > cap && max_batch > 2) max_batch = (max_batch + 1) / 2;
            while ((size_t)lanes * (size_t)max_batch > cap && lanes > 1) lanes--;
            if (d == 0 || total_b / 4 < pool_budget) pool_budget = total_b / 4;
        }
        if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    }
    std::mutex mu;
    int first_rc = ZKHIP_OK, first_job = n_jobs, ctx_rc = ZKHIP_OK;
    std::string first_msg, ctx_msg;
    auto note = [&](int job, int rc, const std::string& msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (job < first_job) { first_job = job; first_rc = rc; first_msg = msg; }
    };
    // batches per device, in job order
    std::vector<std::vector<std::vector<int>>> batches((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) {
        std::vector<std::pair<int, std::vector<int>>> groups;
        for (int i = d; i < n_jobs; i += n_devices) {
            size_t g = 0;
            while (g < groups.size() && groups[g].first != shape[i]) g++;
            if (g == groups.size()) groups.emplace_back(shape[i], std::vector<int>());
            groups[g].second.push_back(i);
        }
        for (auto& g : groups) {
            // equal cuts: sixty-four jobs on two lanes are two batches of thirty-two, not one of max_batch and a remainder
            const size_t n = g.second.size();
            size_t cuts = (n + (size_t)max_batch - 1) / (size_t)max_batch;
            if (cuts < (size_t)lanes && n >= 2 * (size_t)lanes) cuts = (size_t)lanes;
            for (size_t c = 0; c < cuts; c++) {
                const size_t lo = n * c / cuts, hi = n * (c + 1) / cuts;
                if (hi > lo) batches[(size_t)d].emplace_back(g.second.begin() + (long)lo, g.second.begin() + (long)hi);
            }
        }
    }
    std::vector<std::atomic<int>> next((size_t)n_devices);
    for (auto& a : next) a.store(0);
    auto lane = [&](int slot) {
        const int device = devices[slot];
        for (;;) {
            const int k = next[(size_t)slot].fetch_add(1);
            if (k >= (int)batches[(size_t)slot].size()) break;
            const std::vector<int>& jobs = batches[(size_t)slot][(size_t)k];
            const int B = (int)jobs.size();
            std::vector<zkhip_ctx*> ctxs;
            for (int b = 0; b < B; b++) {
                zkhip_ctx* c = pool_take(device);
                if (!c && zkhip_ctx_create(device, nullptr, &c) != ZKHIP_OK) {
                    std::lock_guard<std::mutex> lk(mu);
                    ctx_rc = ZKHIP_ERR_HIP; ctx_msg = zkhip_last_error();
                    break;
                }
                ctxs.push_back(c);
            }
            if (ctxs.empty()) continue;                          // no context at all: the jobs stay unrun
            // fewer contexts than jobs (memory): the batch runs in rounds of ctxs.size()
            const int W = (int)ctxs.size();
            std::vector<hipStream_t> own((size_t)W);
            for (int b = 0; b < W; b++) own[(size_t)b] = ctxs[(size_t)b]->stream;
            const hipStream_t shared = own[0];
            std::vector<char> healthy((size_t)W, 1);
            std::vector<int> retry;                              // jobs whose member ran out of device memory: once more, alone, at the end
            for (int at = 0; at < B; at += W) {
                const int n = B - at < W ? B - at : W;
                if (n == 1) {                                    // nothing to merge with
                    const int i = jobs[(size_t)at];
                    const int rc = run(ctxs[0], i);
                    ran[(size_t)i] = 1;
                    if (rc == ZKHIP_ERR_NOMEM) { retry.push_back(i); healthy[0] = 0; }
                    else if (rc != ZKHIP_OK) { note(i, rc, zkhip_last_error()); healthy[0] = 0; }
                    continue;
                }
                for (int b = 0; b < n; b++) ctxs[(size_t)b]->stream = shared;
                {
                    LaunchBatcher lb(n, shared);
                    auto member = [&](int b) {
                        const int i = jobs[(size_t)(at + b)];
                        // a host-side exception inside ONE member (bad_alloc from a prover's vectors ...) is that member's status with its own
                        // text -- out of memory takes the retry path below --, not a failed batch: the other members' proofs stay good
                        int rc;
                        std::string msg;
                        try { rc = run(ctxs[(size_t)b], i); if (rc != ZKHIP_OK) msg = zkhip_last_error(); }
                        catch (const std::bad_alloc&) { rc = ZKHIP_ERR_NOMEM; msg = "lock-step member: out of host memory"; }
                        catch (const std::exception& e) { rc = ZKHIP_ERR_INTERNAL; msg = std::string("lock-step member: ") + e.what(); }
                        catch (...) { rc = ZKHIP_ERR_INTERNAL; msg = "lock-step member: unknown exception"; }
                        ran[(size_t)i] = 1;
                        if (rc == ZKHIP_ERR_NOMEM) { retry.push_back(i); healthy[(size_t)b] = 0; }      // (fibers of ONE thread: no lock)
                        else if (rc != ZKHIP_OK) { note(i, rc, msg); healthy[(size_t)b] = 0; }
                    };
                    if (lb.ok()) lb.run(member);                 // the members as fibers of this thread, their launches merged
                    else for (int b = 0; b < n; b++) member(b);  // (no pinned memory / stacks: one after the other, unmerged)
                    if (hipStreamSynchronize(shared) != hipSuccess) { (void)hipGetLastError(); for (int b = 0; b < n; b++) healthy[(size_t)b] = 0; }
                    if (lb.failed()) {                            // a merged launch failed: every proof of the batch is suspect
                        for (int b = 0; b < n; b++) {
                            healthy[(size_t)b] = 0;
                            note(jobs[(size_t)(at + b)], ZKHIP_ERR_HIP, "lock-step batch: a merged kernel launch failed");
                        }
                    }
                }
                for (int b = 0; b < n; b++) ctxs[(size_t)b]->stream = own[(size_t)b];
            }
            zkhip_ctx* solo = nullptr;                           // a healthy context stays with this lane while jobs wait for their second try
            for (int b = 0; b < W; b++) {
                if (healthy[(size_t)b] && zkhip_ctx_sync(ctxs[(size_t)b]) == ZKHIP_OK) {
                    if (!retry.empty() && !solo) solo = ctxs[(size_t)b];
                    else pool_give(device, ctxs[(size_t)b]);
                } else zkhip_ctx_destroy(ctxs[(size_t)b]);       // (frees the workspaces of the members that failed)
            }
            if (!retry.empty()) {
                // out of memory inside a member: the job is not lost -- the idle contexts of the pool give their memory back and the job
                // runs once more without a batch around it
                pool_trim(device, 0, 0);
                if (!solo && zkhip_ctx_create(device, nullptr, &solo) != ZKHIP_OK) solo = nullptr;
                for (int i : retry) {
                    const int rc = solo ? run(solo, i) : ZKHIP_ERR_NOMEM;
                    if (rc != ZKHIP_OK) {
                        note(i, rc, solo ? zkhip_last_error() : "lock-step batch: out of device memory, and no context for a second try");
                        if (solo) { zkhip_ctx_destroy(solo); solo = nullptr; }
                        if (zkhip_ctx_create(device, nullptr, &solo) != ZKHIP_OK) solo = nullptr;
                    }
                }
                if (solo) { if (zkhip_ctx_sync(solo) == ZKHIP_OK) pool_give(device, solo); else zkhip_ctx_destroy(solo); }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int slot = 0; slot < n_devices; slot++) {
        const int nb = (int)batches[(size_t)slot].size();
        const int workers = nb < lanes ? nb : lanes;
        for (int t = 0; t < workers; t++) {
            try { pool.emplace_back(lane, slot); }
            catch (...) { if (t == 0) lane(slot); break; }
        }
    }
    for (auto& t : pool) t.join();
    // what the call leaves in the pool: at most lanes x max_batch contexts per device and a quarter of the device's memory
    for (int d = 0; d < n_devices; d++) pool_trim(devices[d], pool_budget, (size_t)lanes * (size_t)max_batch);
    if (first_rc != ZKHIP_OK) { set_error(first_msg); return first_rc; }
    for (int i = 0; i < n_jobs; i++)
        if (!ran[(size_t)i]) { set_error(ctx_msg.empty() ? "prove (lock-step): job not run" : ctx_msg); return ctx_rc != ZKHIP_OK ? ctx_rc : ZKHIP_ERR_INVALID; }
    return ZKHIP_OK;
}
// the device list of a batch entry: NULL (with n_devices == 0) = every visible device; ordinals non-negative and distinct.
// ZKHIP_ERR_NO_DEVICE when nothing is visible (the caller marks its jobs)
int resolve_devices(const int* devices, int n_devices, const char* what, std::vector<int>& devs) {
    devs.clear();
    if (!devices) {
        if (n_devices != 0) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": n_devices must be 0 when devices is NULL (all visible devices)");
        const int n = zkhip_device_count();
        if (n <= 0) return fail(ZKHIP_ERR_NO_DEVICE, "no HIP device visible: libzkhip has no CPU fallback");
        for (int d = 0; d < n; d++) devs.push_back(d);
        return ZKHIP_OK;
    }
    if (n_devices < 1 || n_devices > 64) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": 1..64 devices");
    for (int d = 0; d < n_devices; d++) {
        if (devices[d] < 0) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": negative device ordinal");
        for (int e = 0; e < d; e++) if (devices[e] == devices[d]) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": device listed twice");
        devs.push_back(devices[d]);
    }
    return ZKHIP_OK;
}
}  // namespace zk
}  // extern "C++"

// Job traces are device pointers ON THE DEVICE THE SHARD IS ASSIGNED TO, or host pointers with host_traces.
static int prove_shards_on(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm, int in_flight,
                           int host_traces, const uint32_t* program = nullptr, size_t program_words = 0) {
    for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_INVALID; jobs[i].proof_len = 0; }
#ifdef ZKHIP_AB_HOOKS
    // logical devices (context.cpp): a device trace must have been allocated on the (logical) device its shard is dealt to -- what two
    // physical devices enforce by themselves
    if (logical_devices() > 0 && !host_traces)
        for (int i = 0; i < n_jobs; i++) {
            const int at = logical_device_of(jobs[i].trace), want = zkhip_shard_device(i, devices, n_devices);
            if (at >= 0 && at != want)
                return fail(ZKHIP_ERR_INVALID, "prove_shards: the trace of shard " + std::to_string(i) + " lives on device " + std::to_string(at) +
                                                   ", the shard is dealt to device " + std::to_string(want));
        }
#endif
    std::vector<char> ran;
    auto run = [&](zkhip_ctx* ctx, int i) {
        zkhip_shard_job& j = jobs[i];
#ifdef ZKHIP_AB_HOOKS
        if (logical_devices() > 0 && ctx->logical_device != zkhip_shard_device(i, devices, n_devices)) { j.status = ZKHIP_ERR_INTERNAL; return fail(ZKHIP_ERR_INTERNAL, "prove_shards: a shard ran on a context of another device"); }
#endif
        size_t len = 0;
        int rc;
        if (program)             // every job of the batch is a trace of the same constraint program (device traces)
            rc = zkhip_prove_shard_air(ctx, program, program_words, j.trace, j.ld, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len);
        else
            rc = host_traces
                     ? zkhip_prove_shard_host(ctx, j.trace, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len)
                     : zkhip_prove_shard(ctx, j.trace, j.ld, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len);
        j.status = rc;
        j.proof_len = rc == ZKHIP_OK ? len : 0;
        return rc;
    };
    // a batch of SMALL shards is launch-bound: lock-step lanes (batch.h) when every job is small, one context and stream each otherwise
    const int max_batch = lockstep_batch();
    bool small = max_batch > 1 && n_jobs >= 2 * n_devices;
    std::vector<int> shape((size_t)n_jobs);
    uint64_t max_cells = 0;
    for (int i = 0; i < n_jobs && small; i++) {
        const zkhip_shard_job& j = jobs[i];
        if (j.log_n < 1 || j.log_n > 24 || j.width == 0 || ((uint64_t)j.width << j.log_n) > lockstep_max_cells()) { small = false; break; }
        shape[(size_t)i] = (int)(((uint32_t)j.log_n << 24) ^ j.width);
        const uint64_t cells = ((uint64_t)j.width << j.log_n) << (prm->log_blowup > 1 ? prm->log_blowup - 1 : 0);
        if (cells > max_cells) max_cells = cells;
    }
    if (small) return deal_jobs_lockstep(devices, n_devices, n_jobs, shape.data(), max_batch, lockstep_lanes(), run, ran, max_cells);
    return deal_jobs(devices, n_devices, n_jobs, in_flight, run, ran);
}

int zkhip_prove_shards(int device, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm, int in_flight, int host_traces) {
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards: bad arguments");
    return prove_shards_on(&device, 1, jobs, n_jobs, prm, in_flight, host_traces);
}

int zkhip_shard_device(int shard_index, const int* devices, int n_devices) {
    if (shard_index < 0 || n_devices < 1) return -1;
    return devices ? devices[shard_index % n_devices] : shard_index % n_devices;
}

int zkhip_prove_shards_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm,
                             int in_flight_per_device, int host_traces) {
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards_multi: bad arguments");
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_shards_multi", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_NO_DEVICE; jobs[i].proof_len = 0; }
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    return prove_shards_on(devs.data(), (int)devs.size(), jobs, n_jobs, prm, in_flight_per_device, host_traces);
}

// the same batch when every job is a trace of ONE constraint program (e.g. sixty-four SHA-256 chip traces: sixty-four transcripts)
int zkhip_prove_shards_air_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const uint32_t* program, size_t program_words,
                                 const zkhip_params* prm, int in_flight_per_device) {
    if (!program || program_words < 6) return fail(ZKHIP_ERR_INVALID, "prove_shards_air_multi: null program");
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards_air_multi: bad arguments");
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_shards_air_multi", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_NO_DEVICE; jobs[i].proof_len = 0; }
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    return prove_shards_on(devs.data(), (int)devs.size(), jobs, n_jobs, prm, in_flight_per_device, 0, program, program_words);
}

}  // extern "C"
