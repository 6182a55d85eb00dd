// context.h -- internal definition of zkhip_ctx (host side of libzkhip).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"
#include "../../include/zkhip_hal.h"
#include "babybear.cuh"
#include "kernels.h"

namespace zk {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define ZK_HIP(call)                                            \
    do {                                                        \
        hipError_t _e = (call);                                 \
        if (_e != hipSuccess) return zk::hip_fail(_e, #call);   \
    } while (0)
#define ZK_TRY(call)                 \
    do {                             \
        int _r = (call);             \
        if (_r != ZKHIP_OK) return _r; \
    } while (0)

// twiddle tables of one transform size / shift, built on first use and cached
struct NttPlan {
    int log_n = 0;
    int kind = 0;            // 0 inverse, 1 forward from natural order, 2 forward from transposed coefficients
    uint32_t shift = 0;      // Montgomery; coset shift (kinds 1, 2)
    int m1 = 0, m2 = 0;      // log sizes of the two factors (m1 = 0: single pass)
    uint32_t* pre = nullptr;   // device
    uint32_t* post = nullptr;  // device
    uint32_t* pre_f = nullptr;   // kind 2, 2^20 rows: the same tables in the fused launch's thread order (built on first use)
    uint32_t* post_f = nullptr;
};

// tables of the one-launch LDE of 2^11 .. 2^15 rows (ntt_small.hip): the inter-phase twiddle bases per height, the coset powers per (height, shift)
struct SmallPlan {
    int log_n = 0;
    uint32_t* tw_inv = nullptr;   // device, w_N^(-i), i < N / 32
    uint32_t* tw_fwd = nullptr;   // device, w_N^(+i)
};
struct SmallPre {
    int log_n = 0;
    uint32_t shift = 0;           // Montgomery
    uint32_t* pre = nullptr;      // device, shift^j / N, j < N
};

// radix-R combine twiddles of a 2^21 / 2^22-row transform (ntt.hip, ntt_combine_kernel)
struct BigPlan {
    int log_n = 0;
    int kind = 0;            // 0 inverse (natural groups, carries 1/R); 1 forward, natural groups; 2 forward, bit-reversed groups
    uint32_t shift = 0;      // Montgomery coset shift (kinds 1, 2)
    uint32_t* tw = nullptr;  // device, [R][N / R]
};

// tables of the native column-major passes (2^20-point vectors): pre [1024], post2d [1024][1024]
struct ColPlan {
    int inverse = 0;
    uint32_t shift = 0;      // Montgomery coset shift (forward)
    uint32_t* pre = nullptr;
    uint32_t* post2d = nullptr;
};

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// Roles of the grow-only workspaces of a context (zkhip_ctx::scratch).  One enum for context.cpp and prover.cpp, so that
// no two roles can land on the same slot by accident; lifetimes: a slot belongs to the op that reserved it until that op
// returns, except the S_T* / S_Q* / S_P* / S_FRI_* / S_DINV slots, which live for the whole prove call.
enum Slot {
    S_COEF = 0,        // coefficients of the matrix being extended (op_coset_lde)
    S_TMP,             // bounce buffer of a transform / injected digests of a mixed-height commitment
    S_TLDE, S_TTREE,   // trace LDE + tree (the data group's tree when the code / data split is on)
    S_QCHUNK, S_QLDE, S_QTREE,
    S_DINV, S_PARTIAL, S_OPEN_OUT,
    S_APOW_Q, S_APOW_F,
    S_FRI_LAYERS, S_FRI_TREES,
    S_GATHER_DESC, S_GATHER_OUT,
    S_PERM, S_PLDE, S_PTREE,
    S_CHAL,            // device challenger + per-layer betas / roots of the FRI commit phase (part of the HIP-graph key)
    S_COL_A, S_COL_B,  // column-major adapters (zkhip_batch_*_colmajor)
    S_STAGE,           // staged copy of a host / column-major trace (zkhip_prove_shard_host, zkhip_prove_segment)
    S_RO,
    S_EXTRA_A, S_EXTRA_B,   // large-transform (2^21, 2^22 rows) bounce buffers
    S_CTREE,           // tree of the code group (zkhip_params.code_width)
    S_CHIP,            // trace of a built-in chip (sha256_chip.hip)
    S_LOOKUP,          // machine mode: a chip's interaction records + lookup weights
    S_ADDEND,          // machine mode: the folded lookup constraints of a chip on its quotient domain
    S_CHIP_B,          // second table of a built-in machine (the SHA-256 machine's range table: values at setup, multiplicities per proof)
    S_KEYTRACE,        // keyed machine: a chip's trace rows [preprocessed | main] for its permutation trace
    S_REC_A, S_REC_B, S_REC_C, S_REC_D, S_REC_E, S_REC_F, S_REC_G, S_REC_H, S_REC_I, S_REC_J,   // recursion machines (fri_chip.hip, shard_verifier.inl): traces and tables of their chips, alive for the whole prove call
    S_WIT_A, S_WIT_B, S_WIT_C,   // recursion machines, device witnesses (machine_verifier.inl): the inner proofs' words; plan + per-proof values + scratch; every Poseidon2 row's (state, bit, KP)
    S_COUNT
};

}  // namespace zk

struct zkhip_ctx {
    int device = 0;                      // physical HIP ordinal
    int logical_device = 0;              // the ordinal the caller named (== device except under the A/B build's ZKHIP_LOGICAL_DEVICES)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t* w1024_fwd = nullptr;
    uint32_t* w1024_inv = nullptr;
    uint32_t* w1024f_fwd = nullptr;      // w_1024^(+-u k1) in the fused launch's thread order
    uint32_t* w1024f_inv = nullptr;
    std::deque<zk::NttPlan> plans;   // deque: references stay valid on push_back
    std::deque<zk::BigPlan> big_plans;
    std::deque<zk::SmallPlan> small_plans;
    std::deque<zk::SmallPre> small_pres;
    std::deque<zk::ColPlan> col_plans;
    zk::DeviceBuffer scratch[zk::S_COUNT];   // grow-only workspaces, indexed by zk::Slot
    zkhip_prove_debug debug{};
    bool lde_fusion = true;              // zkhip_ctx_set_lde_fusion: fused middle launch of 2^20-row LDEs (ntt_fused.hip)
    // domain tables (prover.cpp): the current set, and every set built so far (a multi-chip shard switches between
    // the sets of its chips' heights; sets are small and kept until the context goes away)
    struct DomainSet { int log_n, log_blowup; uint32_t *xs, *sel_first, *sel_last, *itw; };
    std::vector<DomainSet> domains;
    int dom_log_n = -1, dom_log_blowup = 0;
    uint32_t* dom_xs = nullptr;        // x_p = g * w_M^bitrev(p), p < M = 2^(log_n + log_blowup); selectors: p < 2N
    uint32_t* dom_sel_first = nullptr; // Z_H(x_p) / (x_p - 1)
    uint32_t* dom_sel_last = nullptr;  // Z_H(x_p) / (x_p - w_N^-1)
    uint32_t* dom_itw = nullptr;       // w_M^-bitrev(i) / 2, i < M / 2 (FRI fold)
    // FRI commit phase as a HIP graph (prover.cpp): ~250 small launches of a fixed shape, replayed from one graph launch;
    // the key is every size and pointer the launches depend on
    std::vector<uint64_t> fri_graph_key;
    hipGraphExec_t fri_graph_exec = nullptr;
    // the SHA-256 machine's proving key made on this context by a batch entry (zkhip_prove_transcripts): kept with the pooled context
    zkhip_machine_key* sha_key = nullptr;
    int sha_key_blowup = 0;
    uint32_t sha_vk[8] = {0};
    // the shard verifier machine's proving key made on this context by zkhip_prove_shard_verifier_batch, with what it is a function of
    zkhip_machine_key* rec_key = nullptr;
    std::vector<uint64_t> rec_key_sig;
    uint32_t rec_vk[8] = {0};
    // grow-only PINNED host block (ctx_host_pinned): work lists that a prover fills on host threads and uploads in one DMA
    void* host_pinned = nullptr;
    size_t host_pinned_bytes = 0;
};

namespace zk {
constexpr int MAX_LOG_ROWS = 22;     // tallest matrix the transforms / the shard prover take (2^20 in two passes; 2^21, 2^22 with a radix-2 / 4 combine pass)
int ctx_reserve(zkhip_ctx* ctx, int slot, size_t bytes, void** out);
int ctx_host_pinned(zkhip_ctx* ctx, size_t bytes, void** out);
// verifier.cpp: a cap on the host threads one verifier call may start (0: none) for callers that are themselves one of many workers,
// and the FRI view of a shard proof WITHOUT hashing its Merkle paths (the caller recomputes every opening and compares the roots)
extern thread_local int t_query_threads_cap;
int fri_view_all_unhashed(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                          uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings, uint32_t* roots, uint32_t* paths,
                          uint32_t transcript[10], const uint32_t* program = nullptr, size_t program_words = 0);      // (program: a version-7 proof of that constraint program)  -- above: the context's pinned host block, grown to `bytes`; contents undefined
int get_plan(zkhip_ctx* ctx, int log_n, int kind, uint32_t shift_monty, const NttPlan** out);
// internal op entry points shared by capi.cpp and prover.cpp (device pointers, ctx stream)
int op_coset_lde(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_ld, uint32_t* d_out, size_t out_ld,
                 int log_n, uint32_t width, int log_blowup, uint32_t shift_monty);
int op_merkle_commit(zkhip_ctx* ctx, const MatDesc* mats, int nmats, int log_h, uint32_t* d_tree);
int op_merkle_commit_mixed(zkhip_ctx* ctx, const MatDesc* mats, const int* log_heights, int nmats, uint32_t* d_tree);
// batch entries (prover.cpp): job i -> devices[i mod n], up to `in_flight` pooled contexts per device, run(ctx, i) -> status
int deal_jobs(const int* devices, int n_devices, int n_jobs, int in_flight, const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran);
// a few host threads for work that should not hold up a prover (checking a proof right after it was made): submit() never blocks,
// wait() returns when everything submitted has run
class HostPool {
public:
    explicit HostPool(int threads);
    ~HostPool();
    void submit(std::function<void()> job);
    void wait();
private:
    struct Impl;
    Impl* impl_;
};
// copies, memsets and waits on the context's stream; inside a lock-step batch (batch.h) their merged forms.  dev_d2h returns with the
// data in dst; dev_h2d returns when src may be reused
int dev_sync(zkhip_ctx* ctx);
int dev_d2h(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes);
int dev_h2d(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes);
int dev_memset(zkhip_ctx* ctx, void* dst, int byte, size_t bytes);
int lockstep_batch();
uint64_t lockstep_max_cells();
int lockstep_lanes();
void lockstep_set(int max_batch, int lanes);
// trace cells of a proof that still counts as small.  Measured as launch-bound (lock-step pays): the 13 KB transcript machine (2^14 x 608 +
// a 2^16-row table = 10 M cells), the query-phase recursion machine (2^15 x 364 + four small chips = 12 M cells).  A proof of 2^26 cells
// (256 MiB of trace) is not launch-bound and takes one context + stream per worker (`in_flight` of them) instead.
constexpr uint64_t LOCKSTEP_MAX_CELLS = (uint64_t)1 << 24;
constexpr size_t LOCKSTEP_BYTES_PER_CELL = 48;               // workspace estimate per trace cell of a member (blowup 2: ~20 measured)
// job_cells: trace cells of the largest job (x 2^(log_blowup - 1)), for the memory budget: lanes x max_batch contexts must fit the device
int deal_jobs_lockstep(const int* devices, int n_devices, int n_jobs, const int* shape, int max_batch, int lanes,
                       const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran, uint64_t job_cells = 0);
int resolve_devices(const int* devices, int n_devices, const char* what, std::vector<int>& devs);
// fri_chip.hip (shard_verifier.inl): the recursion provers' pooled host tables, emptied by zkhip_release_cached_contexts
void rec_release_host_tables();
int physical_device(int device);     // the HIP ordinal behind a device-list entry (identity in the shipped library)
#ifdef ZKHIP_AB_HOOKS
int logical_devices();               // ZKHIP_LOGICAL_DEVICES (0: off)
int logical_device_of(const void* p);   // logical device of the zkhip_malloc allocation that holds p, or -1
#endif
}  // namespace zk
