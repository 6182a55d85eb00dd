// context.cpp -- zkhip_ctx lifecycle, NTT plan cache and the op-level C ABI entry points
// (memory helpers, synthetic data, DFT / coset LDE, Poseidon2 Merkle commit).
// Boundary contract: include/zkhip.h.  No CPU fallback: everything here launches gfx950
// kernels on the context's stream and fails with ZKHIP_ERR_NO_DEVICE without a device.
#include "context.h"
#include "poseidon2.cuh"

#include <atomic>
#include <mutex>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace zk {

extern std::atomic<int> g_live_contexts;       // params.cpp: the Poseidon2 parameter set may only change while this is zero

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
void fiber_tls_swap_error(std::string& err) { g_last_error.swap(err); }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + what;
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? ZKHIP_ERR_NOMEM : ZKHIP_ERR_HIP;
}

int ctx_reserve(zkhip_ctx* ctx, int slot, size_t bytes, void** out) {
    DeviceBuffer& b = ctx->scratch[slot];
    if (b.bytes < bytes) {
        if (b.ptr) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(b.ptr)); b.ptr = nullptr; b.bytes = 0; }
        ZK_HIP(hipMalloc(&b.ptr, bytes));
        b.bytes = bytes;
    }
    *out = b.ptr;
    return ZKHIP_OK;
}

int ctx_host_pinned(zkhip_ctx* ctx, size_t bytes, void** out) {
    if (ctx->host_pinned_bytes < bytes) {
        if (ctx->host_pinned) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipHostFree(ctx->host_pinned)); ctx->host_pinned = nullptr; ctx->host_pinned_bytes = 0; }
        ZK_HIP(hipHostMalloc(&ctx->host_pinned, bytes, hipHostMallocDefault));
        ctx->host_pinned_bytes = bytes;
    }
    *out = ctx->host_pinned;
    return ZKHIP_OK;
}

// factorisation N = M1 * M2 used by every two-pass transform of this size
static void split(int log_n, int* m1, int* m2) {
    if (log_n <= 10) { *m1 = 0; *m2 = log_n; return; }
    *m2 = (log_n + 1) / 2;
    *m1 = log_n - *m2;
}

int get_plan(zkhip_ctx* ctx, int log_n, int kind, uint32_t shift, const NttPlan** out) {
    for (const NttPlan& p : ctx->plans)
        if (p.log_n == log_n && p.kind == kind && (kind == 0 || p.shift == shift)) { *out = &p; return ZKHIP_OK; }
    NttPlan p;
    p.log_n = log_n; p.kind = kind; p.shift = shift;
    split(log_n, &p.m1, &p.m2);
    const size_t n = (size_t)1 << log_n;
    const uint32_t w = two_adic_generator(log_n);
    hipStream_t s = ctx->stream;
    if (kind == 0) {
        // inverse: post[i1*M2 + k2] = w^-(i1 k2) / N   (single pass: post[k] = 1/N)
        const uint32_t ninv = finv(to_monty((uint32_t)(n % P)));
        ZK_HIP(hipMalloc((void**)&p.post, n * 4));
        if (p.m1 == 0) ZK_HIP(launch_pow_table(p.post, n, MONTY_R1, ninv, s));
        else ZK_HIP(launch_post_table(p.post, 1u << p.m1, 1u << p.m2, finv(w), MONTY_R1, ninv, s));
    } else if (kind == 1) {
        // forward from natural order, coset shift s: x_j * s^j
        //   two passes: j = i1 + M1 i2; pre[i2] = (s^M1)^i2; post[i1*M2 + k2] = w^(i1 k2) s^i1
        //   one pass:   pre[j] = s^j
        if (p.m1 == 0) {
            if (shift != MONTY_R1) {
                ZK_HIP(hipMalloc((void**)&p.pre, n * 4));
                ZK_HIP(launch_pow_table(p.pre, n, shift, MONTY_R1, s));
            }
        } else {
            const uint32_t M1 = 1u << p.m1, M2 = 1u << p.m2;
            if (shift != MONTY_R1) {
                ZK_HIP(hipMalloc((void**)&p.pre, (size_t)M2 * 4));
                ZK_HIP(launch_pow_table(p.pre, M2, fpow(shift, M1), MONTY_R1, s));
            }
            ZK_HIP(hipMalloc((void**)&p.post, n * 4));
            ZK_HIP(launch_post_table(p.post, M1, M2, w, shift, MONTY_R1, s));
        }
    } else {
        // forward from TRANSPOSED coefficients (c[M2 k1 + k2] at row k2*M1 + k1):
        //   j = i1' + M1' i2' with M1' = M2 (block index), M2' = M1
        //   pre[i2'] = (s^M1')^i2';  post[i1'*M2' + k2'] = w^(i1' k2') s^i1'
        if (p.m1 == 0) return fail(ZKHIP_ERR_INTERNAL, "transposed plan needs two passes");
        const uint32_t M1p = 1u << p.m2, M2p = 1u << p.m1;
        ZK_HIP(hipMalloc((void**)&p.pre, (size_t)M2p * 4));
        ZK_HIP(launch_pow_table(p.pre, M2p, fpow(shift, M1p), MONTY_R1, s));
        ZK_HIP(hipMalloc((void**)&p.post, n * 4));
        ZK_HIP(launch_post_table(p.post, M1p, M2p, w, shift, MONTY_R1, s));
    }
    ctx->plans.push_back(p);
    *out = &ctx->plans.back();
    return ZKHIP_OK;
}

enum { LDE_I1 = 0, LDE_I2 = 1, LDE_F1 = 2, LDE_F2 = 3, LDE_I1_BLOCKS = 4 };
constexpr int ZKHIP_ERR_UNSUPPORTED_FUSION = -100;     // private to this file: lde_fused_args -> lde_two_pass (take the four-pass sequence)

static NttPassArgs base_args(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld,
                             uint32_t width, bool inverse) {
    NttPassArgs a{};
    a.in = in; a.out = out; a.in_ld = in_ld; a.out_ld = out_ld; a.ncols = width;
    a.w1024 = inverse ? ctx->w1024_inv : ctx->w1024_fwd;
    a.map_mode = 255;
#ifdef ZKHIP_AB_HOOKS
    // A/B builds only (libzkhip_ab.so, `make ab`; tools/): tuning knobs read from the environment.  The shipped library is
    // compiled without them: it reads no environment variable and its launches cannot be altered from outside.
    static const int force_cpt = [] { const char* e = getenv("ZKHIP_NTT_CPT"); return e ? atoi(e) : 0; }();
    static const bool has_map = getenv("ZKHIP_NTT_MAP") != nullptr;      // re-read per launch when present at start-up
    int map_mode = 255;                                                   // automatic (launch_ntt_pass)
    if (has_map) { const char* e = getenv("ZKHIP_NTT_MAP"); map_mode = e ? atoi(e) : 255; }
    static const int fast = [] { const char* e = getenv("ZKHIP_NTT_FAST"); return e ? atoi(e) : 4; }();
    // ZKHIP_NTT_FAST: unset/4/0 = tile-per-workgroup kernel, 1 = persistent 1024 x 32 kernel;
    // ZKHIP_NTT_CPT: 1 / 2 columns per lane (unset: 2 for 1024-row tiles of an even, 8-byte aligned shape)
    a.fast_path = fast == 4 ? 0u : (fast == 0 ? 2u : (uint32_t)fast);
    static const bool has_dbg = getenv("ZKHIP_NTT_DEBUG") != nullptr;
    int dbg = 0;
    if (has_dbg) { const char* e = getenv("ZKHIP_NTT_DEBUG"); dbg = e ? atoi(e) : 0; }
    a.debug_flags = (uint32_t)dbg;
    a.cols_per_thread = (uint32_t)force_cpt;
    a.map_mode = (uint32_t)map_mode;
    static const bool has_perm = getenv("ZKHIP_NTT_PERM") != nullptr;
    if (has_perm) { const char* e = getenv("ZKHIP_NTT_PERM"); a.tile_perm = e ? strtoull(e, nullptr, 16) : 0; }
#endif
    return a;
}

// inverse DFT, natural in; coefficients out natural (transposed = false) or in the
// transposed order c[M2 k1 + k2] -> row k2*M1 + k1 (transposed = true, two-pass sizes)
static int run_inverse(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld,
                       int log_n, uint32_t width, bool transposed) {
    const NttPlan* p;
    ZK_TRY(get_plan(ctx, log_n, 0, 0, &p));
    if (p->m1 == 0) {
        NttPassArgs a = base_args(ctx, in, in_ld, out, out_ld, width, true);
        a.num_tiles = 1; a.log_m = log_n; a.in_tile_mul = 0; a.in_stride = 1; a.out_tile_mul = 0; a.out_stride = 1;
        a.post = p->post;
        ZK_HIP(launch_ntt_pass(a, true, ctx->stream));
        return ZKHIP_OK;
    }
    const uint64_t M1 = 1ull << p->m1, M2 = 1ull << p->m2;
    NttPassArgs a = base_args(ctx, in, in_ld, out, out_ld, width, true);
    a.num_tiles = (uint32_t)M1; a.log_m = p->m2;
    a.in_tile_mul = 1; a.in_stride = M1; a.out_tile_mul = 1; a.out_stride = M1; a.post = p->post;
    ZK_HIP(launch_ntt_pass(a, true, ctx->stream));
    NttPassArgs b = base_args(ctx, out, out_ld, out, out_ld, width, true);
    b.num_tiles = (uint32_t)M2; b.log_m = p->m1;
    b.in_tile_mul = M1; b.in_stride = 1;
    if (transposed) { b.out_tile_mul = M1; b.out_stride = 1; }
    else { b.out_tile_mul = 1; b.out_stride = M2; }
    if (!transposed) {
        // natural output rows interleave across tiles: not in-place safe -> bounce
        void* tmp;
        ZK_TRY(ctx_reserve(ctx, S_TMP, ((size_t)1 << log_n) * width * 4, &tmp));
        b.out = (uint32_t*)tmp; b.out_ld = width;
        ZK_HIP(launch_ntt_pass(b, true, ctx->stream));
        ZK_HIP(hipMemcpy2DAsync(out, out_ld * 4, tmp, (size_t)width * 4, (size_t)width * 4, (size_t)1 << log_n,
                                hipMemcpyDeviceToDevice, ctx->stream));
        return ZKHIP_OK;
    }
    ZK_HIP(launch_ntt_pass(b, true, ctx->stream));
    return ZKHIP_OK;
}

// forward coset DFT of natural-order input; output natural or bit-reversed rows
static int run_forward_natural(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld,
                               int log_n, uint32_t width, uint32_t shift, bool bitrev_out) {
    const NttPlan* p;
    ZK_TRY(get_plan(ctx, log_n, 1, shift, &p));
    if (p->m1 == 0) {
        NttPassArgs a = base_args(ctx, in, in_ld, out, out_ld, width, false);
        a.num_tiles = 1; a.log_m = log_n; a.in_tile_mul = 0; a.in_stride = 1; a.out_tile_mul = 0; a.out_stride = 1;
        a.pre = p->pre; a.bitrev_out = bitrev_out ? 1 : 0;
        ZK_HIP(launch_ntt_pass(a, false, ctx->stream));
        return ZKHIP_OK;
    }
    const uint64_t M1 = 1ull << p->m1, M2 = 1ull << p->m2;
    // pass 1 goes through a scratch matrix so that `in` is preserved and `out` may be strided
    void* tmp;
    ZK_TRY(ctx_reserve(ctx, S_TMP, ((size_t)1 << log_n) * width * 4, &tmp));
    NttPassArgs a = base_args(ctx, in, in_ld, (uint32_t*)tmp, width, width, false);
    a.num_tiles = (uint32_t)M1; a.log_m = p->m2;
    a.in_tile_mul = 1; a.in_stride = M1; a.out_tile_mul = 1; a.out_stride = M1;
    a.pre = p->pre; a.post = p->post; a.bitrev_out = bitrev_out ? 1 : 0;
    ZK_HIP(launch_ntt_pass(a, false, ctx->stream));
    NttPassArgs b = base_args(ctx, (uint32_t*)tmp, width, out, out_ld, width, false);
    b.num_tiles = (uint32_t)M2; b.log_m = p->m1;
    b.in_tile_mul = M1; b.in_stride = 1;
    if (bitrev_out) { b.out_tile_mul = M1; b.out_stride = 1; b.bitrev_out = 1; }
    else { b.out_tile_mul = 1; b.out_stride = M2; b.bitrev_out = 0; }
    ZK_HIP(launch_ntt_pass(b, false, ctx->stream));
    return ZKHIP_OK;
}

// The four launches of a two-pass LDE (log_n >= 11), built in ONE place: op_coset_lde enqueues them and zkhip_ntt_pass
// (which = 2..5) replays any one of them on the context's own workspaces for the roofline measurement.
//   I1  inverse, strided in -> strided out (trace -> coefficient workspace), post = w^-(i1 k2) / N
//   I2  inverse, contiguous tiles in place on the coefficient workspace (coefficients left in transposed order)
//   F1  forward, block in (transposed coefficients) -> strided bit-reversed out, pre = (s^M1')^i2', post = w^(i1' k2') s^i1'
//   F2  forward, contiguous tiles in place on the LDE, bit-reversed inside the tile
int lde_pass_args(zkhip_ctx* ctx, int which, const uint32_t* in, size_t in_ld, uint32_t* coef, size_t coef_ld, uint32_t* dst, size_t out_ld,
                  int log_n, uint32_t width, uint32_t coset_shift, NttPassArgs* out, bool* inverse) {
    int m1, m2;
    split(log_n, &m1, &m2);
    if (m1 == 0) return fail(ZKHIP_ERR_INTERNAL, "lde_pass_args: single-pass size");
    const uint64_t M1 = 1ull << m1, M2 = 1ull << m2;
    const NttPlan* p;
    if (which == LDE_I1 || which == LDE_I2 || which == LDE_I1_BLOCKS) {
        ZK_TRY(get_plan(ctx, log_n, 0, 0, &p));
        *inverse = true;
        if (which == LDE_I1 || which == LDE_I1_BLOCKS) {
            NttPassArgs a = base_args(ctx, in, in_ld, coef, coef_ld, width, true);
            a.num_tiles = (uint32_t)M1; a.log_m = (uint32_t)m2;
            a.in_tile_mul = 1; a.in_stride = M1; a.out_tile_mul = 1; a.out_stride = M1; a.post = p->post;
            // in front of the fused middle launch a tile leaves as ONE contiguous block of the (private) workspace: the strided
            // side of the transposition moves into the fused launch, which has the slack for it
            if (which == LDE_I1_BLOCKS) { a.out_tile_mul = M2; a.out_stride = 1; }
            *out = a;
        } else {
            NttPassArgs b = base_args(ctx, coef, coef_ld, coef, coef_ld, width, true);
            b.num_tiles = (uint32_t)M2; b.log_m = (uint32_t)m1;
            b.in_tile_mul = M1; b.in_stride = 1; b.out_tile_mul = M1; b.out_stride = 1;
            *out = b;
        }
        return ZKHIP_OK;
    }
    ZK_TRY(get_plan(ctx, log_n, 2, coset_shift, &p));
    *inverse = false;
    const uint64_t M1p = M2, M2p = M1;                               // forward factors (swapped)
    if (which == LDE_F1) {
        NttPassArgs a = base_args(ctx, coef, coef_ld, dst, out_ld, width, false);
        a.num_tiles = (uint32_t)M1p; a.log_m = (uint32_t)m1;
        a.in_tile_mul = M2p; a.in_stride = 1;
        a.out_tile_mul = 1; a.out_stride = M1p; a.bitrev_out = 1;
        a.pre = p->pre; a.post = p->post;
        *out = a;
    } else {
        NttPassArgs b = base_args(ctx, dst, out_ld, dst, out_ld, width, false);
        b.num_tiles = (uint32_t)M2p; b.log_m = (uint32_t)m2;
        b.in_tile_mul = M1p; b.in_stride = 1; b.out_tile_mul = M1p; b.out_stride = 1; b.bitrev_out = 1;
        *out = b;
    }
    return ZKHIP_OK;
}

// The middle of a 2^20-row LDE as one launch (ntt_fused.hip): second inverse pass + first forward pass of two cosets, reading the
// blocks LDE_I1_BLOCKS left (tile k1 = rows n2 * 1024 + k1) and writing what LDE_F1 would have written for each coset.
static bool lde_fused_shape(int log_n, uint32_t width, size_t coef_ld, size_t out_ld, int cosets) {
    return log_n == 20 && width % 32 == 0 && coef_ld % 2 == 0 && out_ld % 2 == 0 && cosets >= 2 && cosets % 2 == 0;
}
int lde_fused_args(zkhip_ctx* ctx, const uint32_t* coef, size_t coef_ld, uint32_t* const* dsts, size_t out_ld, int log_n, uint32_t width,
                   const uint32_t* coset_shifts, LdeFusedArgs* out) {
    int m1, m2;
    split(log_n, &m1, &m2);
    if (m1 != 10 || m2 != 10) return fail(ZKHIP_ERR_INTERNAL, "lde_fused_args: 1024 x 1024 transforms only");
    LdeFusedArgs f{};
    f.in = coef; f.in_ld = coef_ld; f.out_ld = out_ld; f.ncols = width; f.num_tiles = 1024;
    f.in_tile_mul = 1; f.in_stride = 1024; f.out_tile_mul = 1; f.out_stride = 1024;
    f.w1024_inv = ctx->w1024f_inv; f.w1024_fwd = ctx->w1024f_fwd;
    for (int t = 0; t < FUSED_COSETS; t++) {
        const NttPlan* p;
        ZK_TRY(get_plan(ctx, log_n, 2, coset_shifts[t], &p));
        if (!p->post_f) {
            NttPlan* mp = const_cast<NttPlan*>(p);       // plans live in the context's deque; the thread-order copies are filled in lazily
            uint32_t *a = nullptr, *b = nullptr;
            ZK_HIP(hipMalloc((void**)&a, 1024 * 4));
            if (hipMalloc((void**)&b, (size_t)1024 * 1024 * 4) != hipSuccess) { (void)hipFree(a); return fail(ZKHIP_ERR_HIP, "hipMalloc (fused tables)"); }
            // the plan owns the tables only once BOTH are filled: a failed launch must not leave pointers to garbage behind
            hipError_t e = launch_fused_table(a, p->pre, 1, 1, ctx->stream);
            if (e == hipSuccess) e = launch_fused_table(b, p->post, 1024, 2, ctx->stream);
            if (e != hipSuccess) { (void)hipFree(a); (void)hipFree(b); return hip_fail(e, "fused tables"); }
            mp->pre_f = a; mp->post_f = b;
        }
        f.out[t] = dsts[t]; f.pre[t] = p->pre_f; f.post[t] = p->post_f;
    }
#ifdef ZKHIP_AB_HOOKS
    static const bool has_rot = getenv("ZKHIP_FUSED_ROT") != nullptr, has_grid = getenv("ZKHIP_FUSED_GRID") != nullptr;
    if (has_rot) { const char* e = getenv("ZKHIP_FUSED_ROT"); f.map_rot = e ? (uint32_t)atoi(e) : 0u; }
    if (has_grid) { const char* e = getenv("ZKHIP_FUSED_GRID"); f.grid = e ? (uint32_t)atoi(e) : 0u; }
#endif
    if (!lde_fused_supported(f)) return fail(ZKHIP_ERR_UNSUPPORTED_FUSION, "lde_fused_args: unsupported shape");
    *out = f;
    return ZKHIP_OK;
}

// A two-pass LDE (2^11 .. 2^20 rows) of one matrix onto `cosets` destinations with their shifts.  2^20 rows x a multiple of 32
// columns, even coset count: I1 (strided in -> blocks), the fused middle launch per pair of cosets, F2 per coset = 36 B per
// trace cell at blowup 2.  Everything else: I1, I2, and F1 + F2 per coset (48 B).
static int lde_two_pass(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* coef, size_t coef_ld, uint32_t* const* dsts, size_t out_ld,
                        int log_n, uint32_t width, const uint32_t* coset_shifts, int cosets, bool run_f2 = true) {
    NttPassArgs a;
    bool inv;
    bool fused = lde_fused_shape(log_n, width, coef_ld, out_ld, cosets) && ctx->lde_fusion;
    LdeFusedArgs fargs[16 / FUSED_COSETS];
    for (int t = 0; fused && t < cosets; t += FUSED_COSETS) {        // all launches of the middle are built BEFORE the first pass is enqueued:
        const int rc = lde_fused_args(ctx, coef, coef_ld, dsts + t, out_ld, log_n, width, coset_shifts + t, &fargs[t / FUSED_COSETS]);
        if (rc == ZKHIP_ERR_UNSUPPORTED_FUSION) fused = false;       // a shape the fused kernel refuses takes the four-pass sequence, it is not an error
        else if (rc != ZKHIP_OK) return rc;
    }
    if (fused) {
        ZK_TRY(lde_pass_args(ctx, LDE_I1_BLOCKS, in, in_ld, coef, coef_ld, nullptr, 0, log_n, width, 0, &a, &inv));
        ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        for (int t = 0; t < cosets; t += FUSED_COSETS) ZK_HIP(launch_lde_fused(fargs[t / FUSED_COSETS], ctx->stream));
    } else {
        ZK_TRY(lde_pass_args(ctx, LDE_I1, in, in_ld, coef, coef_ld, nullptr, 0, log_n, width, 0, &a, &inv));
        ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        ZK_TRY(lde_pass_args(ctx, LDE_I2, nullptr, 0, coef, coef_ld, nullptr, 0, log_n, width, 0, &a, &inv));
        ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        for (int t = 0; t < cosets; t++) {
            ZK_TRY(lde_pass_args(ctx, LDE_F1, nullptr, 0, coef, coef_ld, dsts[t], out_ld, log_n, width, coset_shifts[t], &a, &inv));
            ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        }
    }
    if (!run_f2) return ZKHIP_OK;
    for (int t = 0; t < cosets; t++) {
        ZK_TRY(lde_pass_args(ctx, LDE_F2, nullptr, 0, coef, coef_ld, dsts[t], out_ld, log_n, width, coset_shifts[t], &a, &inv));
        ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
    }
    return ZKHIP_OK;
}

// ---- transforms of 2^21 and 2^22 rows (SP1 core shards reach 2^21 - 2^22 rows: reference benchmark.md:9).  N = R N', N' = 2^20,
// R = 2 or 4.  Decimation in time over the row classes j = n mod R: the class-j rows form a sub-matrix (first row j, row pitch
// R ld) that the two-pass N'-point machinery above transforms as it stands, and one streaming radix-R pass (ntt_combine_kernel,
// 8 B/element) finishes:   X[k1 + N' k2] = sum_j w_R^(j k2) (s^j w_N^(j k1)) Y_j[k1],   Y_j = coset-DFT_N' of class j, shift s^R.
// The inverse runs the other way round (radix-R step first).  With bit-reversed output the R values of one k1 are R ADJACENT
// rows (R bitrev(k1) + bitrev(k2)), so the combine pass works in place on the LDE.
constexpr int BIG_INNER_LOG = 20;
static int get_big_plan(zkhip_ctx* ctx, int log_n, int kind, uint32_t shift, const BigPlan** out) {
    for (const BigPlan& p : ctx->big_plans)
        if (p.log_n == log_n && p.kind == kind && (kind == 0 || p.shift == shift)) { *out = &p; return ZKHIP_OK; }
    BigPlan p;
    p.log_n = log_n; p.kind = kind; p.shift = shift;
    const uint32_t R = 1u << (log_n - BIG_INNER_LOG);
    const uint64_t groups = (uint64_t)1 << BIG_INNER_LOG;
    const uint32_t w = two_adic_generator(log_n);
    ZK_HIP(hipMalloc((void**)&p.tw, (size_t)R * groups * 4));
    hipError_t e;
    if (kind == 0) e = launch_combine_table(p.tw, R, groups, finv(w), MONTY_R1, finv(to_monty(R)), 0, ctx->stream);
    else e = launch_combine_table(p.tw, R, groups, w, shift, MONTY_R1, kind == 2 ? BIG_INNER_LOG : 0, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p.tw); return hip_fail(e, "combine_table"); }
    ctx->big_plans.push_back(p);
    *out = &ctx->big_plans.back();
    return ZKHIP_OK;
}
// the sub-matrix passes address a tile through one 32-bit buffer offset: (M - 1) * stride * (R ld) * 4 bytes must stay below 4 GiB
static int check_big_width(int log_n, size_t ld_a, size_t ld_b, const char* what) {
    const size_t R = (size_t)1 << (log_n - BIG_INNER_LOG);
    const size_t ld = ld_a > ld_b ? ld_a : ld_b;
    if (4ull * (1023ull * 1024ull * R * ld + 1024ull) >= (1ull << 32))
        return fail(ZKHIP_ERR_INVALID, std::string(what) + ": 2^21-row matrices take a row pitch of at most 512 words, 2^22-row matrices 256");
    return ZKHIP_OK;
}
static CombineArgs combine_args(const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld, uint32_t width, int log_n, const BigPlan* p) {
    CombineArgs c{};
    c.in = in; c.in_ld = in_ld; c.out = out; c.out_ld = out_ld; c.ncols = width;
    c.groups = (uint64_t)1 << BIG_INNER_LOG; c.log_r = log_n - BIG_INNER_LOG; c.tw = p->tw;
    return c;
}
static int run_inverse_big(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld, int log_n, uint32_t width) {
    ZK_TRY(check_big_width(log_n, width, out_ld, "dft"));
    const uint64_t R = 1ull << (log_n - BIG_INNER_LOG), Np = 1ull << BIG_INNER_LOG;
    const BigPlan* bp;
    ZK_TRY(get_big_plan(ctx, log_n, 0, 0, &bp));
    void* tmp;
    ZK_TRY(ctx_reserve(ctx, S_EXTRA_A, ((size_t)1 << log_n) * width * 4, &tmp));
    CombineArgs c = combine_args(in, in_ld, (uint32_t*)tmp, width, width, log_n, bp);
    // the classes leave the radix-R step CLASS-MAJOR (class j = rows [j N', (j + 1) N') of tmp, pitch = width): the strided pass that
    // reads them then strides by 1024 dense rows, as at 2^20, not by 1024 R rows (2 MiB and more: 0.55 instead of 0.46 ms per pass)
    c.in_group_mul = 1; c.in_elem_mul = Np; c.out_group_mul = 1; c.out_elem_mul = Np; c.inverse = 1;
    ZK_HIP(launch_ntt_combine(c, ctx->stream));
    for (uint64_t j = 0; j < R; j++)
        ZK_TRY(run_inverse(ctx, (const uint32_t*)tmp + j * Np * width, width, out + j * out_ld, R * out_ld, BIG_INNER_LOG, width, false));
    return ZKHIP_OK;
}
static int run_forward_natural_big(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld, int log_n, uint32_t width,
                                   uint32_t shift, bool bitrev_out) {
    ZK_TRY(check_big_width(log_n, in_ld, bitrev_out ? out_ld : width, "dft"));
    const uint64_t R = 1ull << (log_n - BIG_INNER_LOG), Np = 1ull << BIG_INNER_LOG;
    const uint32_t sR = fpow(shift, R);
    const BigPlan* bp;
    ZK_TRY(get_big_plan(ctx, log_n, bitrev_out ? 2 : 1, shift, &bp));
    if (bitrev_out) {
        for (uint64_t j = 0; j < R; j++)
            ZK_TRY(run_forward_natural(ctx, in + j * in_ld, R * in_ld, out + j * out_ld, R * out_ld, BIG_INNER_LOG, width, sR, true));
        CombineArgs c = combine_args(out, out_ld, out, out_ld, width, log_n, bp);
        c.in_group_mul = R; c.in_elem_mul = 1; c.out_group_mul = R; c.out_elem_mul = 1; c.bitrev_out = 1;
        ZK_HIP(launch_ntt_combine(c, ctx->stream));
        return ZKHIP_OK;
    }
    void* tmp;
    ZK_TRY(ctx_reserve(ctx, S_EXTRA_A, ((size_t)1 << log_n) * width * 4, &tmp));
    for (uint64_t j = 0; j < R; j++)
        ZK_TRY(run_forward_natural(ctx, in + j * in_ld, R * in_ld, (uint32_t*)tmp + j * Np * width, width, BIG_INNER_LOG, width, sR, false));
    CombineArgs c = combine_args((const uint32_t*)tmp, width, out, out_ld, width, log_n, bp);
    c.in_group_mul = 1; c.in_elem_mul = Np; c.out_group_mul = 1; c.out_elem_mul = Np;
    ZK_HIP(launch_ntt_combine(c, ctx->stream));
    return ZKHIP_OK;
}
static int coset_lde_big(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld, int log_n, uint32_t width,
                         int log_blowup, uint32_t shift) {
    // blowup 2: every tile pass runs on a dense 2^20-row class (pitch = width; the radix-R passes address 64 bits), so the pitch of the
    // caller's matrices does not matter; other blowups write every R-th row of the LDE
    if (log_blowup == 1) {
        if (4ull * (1023ull * 1024ull * width + 1024ull) >= (1ull << 32)) return fail(ZKHIP_ERR_INVALID, "coset_lde: at most 1024 columns");
    } else ZK_TRY(check_big_width(log_n, width, out_ld, "coset_lde"));
    const size_t n = (size_t)1 << log_n;
    const uint64_t R = 1ull << (log_n - BIG_INNER_LOG), Np = 1ull << BIG_INNER_LOG;
    void *coef_v, *tmp;
    ZK_TRY(ctx_reserve(ctx, S_COEF, n * width * 4, &coef_v));
    ZK_TRY(ctx_reserve(ctx, S_EXTRA_A, n * width * 4, &tmp));
    uint32_t* coef = (uint32_t*)coef_v;
    const BigPlan* bp;
    ZK_TRY(get_big_plan(ctx, log_n, 0, 0, &bp));
    CombineArgs c = combine_args(in, in_ld, (uint32_t*)tmp, width, width, log_n, bp);
    c.in_group_mul = 1; c.in_elem_mul = Np; c.out_group_mul = 1; c.out_elem_mul = Np; c.inverse = 1;      // class-major, as in run_inverse_big
    ZK_HIP(launch_ntt_combine(c, ctx->stream));
    NttPassArgs a;
    bool inv;
    const int B = 1 << log_blowup;
    const uint32_t wnb = two_adic_generator(log_n + log_blowup);
    uint32_t st_all[16], stR_all[16];
    uint32_t* dst_all[16];
    for (int t = 0; t < B; t++) {
        st_all[t] = fmul(shift, fpow(wnb, (uint64_t)t));
        stR_all[t] = fpow(st_all[t], R);
        dst_all[t] = out + (size_t)reverse_bits((uint32_t)t, log_blowup) * n * out_ld;
    }
    // Blowup 2: the cosets stay CLASS-MAJOR (dense 2^20-row matrices) until the radix-R step, which then reads class-major and writes
    // the LDE's adjacent rows: the middle launch and F2 work on dense matrices as at 2^20 instead of on every R-th row of the LDE
    // (1.16 against 1.22 ms per middle launch).  Coset 0 takes over tmp, class by class as the first pass has consumed it; coset 1
    // a second workspace of the trace's size.
    uint32_t* cm[2] = {nullptr, nullptr};
    if (B == 2) {
        void* second;
        ZK_TRY(ctx_reserve(ctx, S_EXTRA_B, n * width * 4, &second));
        cm[0] = (uint32_t*)tmp; cm[1] = (uint32_t*)second;
    }
    for (uint64_t j = 0; j < R; j++) {       // class j: coefficients c[R q + j] -> the first forward pass of every coset (no F2 yet)
        uint32_t* dj[16];
        for (int t = 0; t < B; t++) dj[t] = cm[0] ? cm[t] + j * Np * width : dst_all[t] + j * out_ld;
        ZK_TRY(lde_two_pass(ctx, (const uint32_t*)tmp + j * Np * width, width, coef + j * Np * width, width, dj, cm[0] ? (size_t)width : R * out_ld, BIG_INNER_LOG, width,
                            stR_all, B, /*run_f2=*/false));
    }
    for (int t = 0; t < B; t++) {
        const uint32_t st = st_all[t], stR = stR_all[t];
        uint32_t* dst = dst_all[t];
        for (uint64_t j = 0; j < R; j++) {
            uint32_t* dj = cm[0] ? cm[t] + j * Np * width : dst + j * out_ld;
            ZK_TRY(lde_pass_args(ctx, LDE_F2, nullptr, 0, coef + j * Np * width, width, dj, cm[0] ? (size_t)width : R * out_ld, BIG_INNER_LOG, width, stR, &a, &inv));
            ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        }
        const BigPlan* fp;
        ZK_TRY(get_big_plan(ctx, log_n, 2, st, &fp));
        CombineArgs f = cm[0] ? combine_args(cm[t], width, dst, out_ld, width, log_n, fp) : combine_args(dst, out_ld, dst, out_ld, width, log_n, fp);
        if (cm[0]) { f.in_group_mul = 1; f.in_elem_mul = Np; } else { f.in_group_mul = R; f.in_elem_mul = 1; }
        f.out_group_mul = R; f.out_elem_mul = 1; f.bitrev_out = 1;
        ZK_HIP(launch_ntt_combine(f, ctx->stream));
    }
    return ZKHIP_OK;
}

// ---- 2^11 .. 2^15 rows: the whole LDE as ONE launch (ntt_small.hip), 4 + 4 * 2^b bytes per trace cell.  The tables are per context, built
// on first use on the context's stream: the twiddle bases of a height (N / 32 words each way) and, per (height, coset shift), shift^j / N.
constexpr int SMALL_LDE_MIN_LOG = 11, SMALL_LDE_MAX_LOG = 15, SMALL_LDE_DEFAULT_MAX_LOG = 13;
constexpr uint32_t SMALL_LDE_MIN_WIDTH = 128;
static int get_small_plan(zkhip_ctx* ctx, int log_n, const SmallPlan** out) {
    for (const SmallPlan& p : ctx->small_plans) if (p.log_n == log_n) { *out = &p; return ZKHIP_OK; }
    SmallPlan p;
    p.log_n = log_n;
    const size_t m = (size_t)1 << (log_n - 5);
    const uint32_t w = two_adic_generator(log_n);
    ZK_HIP(hipMalloc((void**)&p.tw_inv, m * 4));
    if (hipMalloc((void**)&p.tw_fwd, m * 4) != hipSuccess) { (void)hipFree(p.tw_inv); return fail(ZKHIP_ERR_HIP, "hipMalloc (small LDE tables)"); }
    hipError_t e = launch_pow_table(p.tw_inv, m, finv(w), MONTY_R1, ctx->stream);
    if (e == hipSuccess) e = launch_pow_table(p.tw_fwd, m, w, MONTY_R1, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p.tw_inv); (void)hipFree(p.tw_fwd); return hip_fail(e, "small LDE tables"); }
    ctx->small_plans.push_back(p);
    *out = &ctx->small_plans.back();
    return ZKHIP_OK;
}
static int get_small_pre(zkhip_ctx* ctx, int log_n, uint32_t shift, const uint32_t** out) {
    for (const SmallPre& p : ctx->small_pres) if (p.log_n == log_n && p.shift == shift) { *out = p.pre; return ZKHIP_OK; }
    SmallPre p;
    p.log_n = log_n; p.shift = shift;
    const size_t n = (size_t)1 << log_n;
    ZK_HIP(hipMalloc((void**)&p.pre, n * 4));
    const hipError_t e = launch_pow_table(p.pre, n, shift, finv(to_monty((uint32_t)n)), ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p.pre); return hip_fail(e, "small LDE coset powers"); }
    ctx->small_pres.push_back(p);
    *out = p.pre;
    return ZKHIP_OK;
}
// ZKHIP_OK with *done = true when the launch was enqueued; *done = false: a shape it does not take (the caller goes on to the pass kernels)
static int lde_small(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld, int log_n, uint32_t width, int log_blowup,
                     uint32_t shift, bool* done) {
    *done = false;
    if (log_n < SMALL_LDE_MIN_LOG || log_n > SMALL_LDE_MAX_LOG || !ctx->lde_fusion) return ZKHIP_OK;
    // where it pays (tools/small_lde_time.py, profiles/r06_small_lde.md): wide matrices of up to 2^13 rows -- 16 .. 64-byte row chunks per workgroup;
    // at 2^14 / 2^15 rows (8 / 4 bytes per row and workgroup, one workgroup per CU with its load and store phases exposed) and for narrow
    // matrices (one or two workgroups: latency of the whole column against six short launches) the pass kernels are faster
    int max_log = SMALL_LDE_DEFAULT_MAX_LOG;
    uint32_t min_width = SMALL_LDE_MIN_WIDTH;
#ifdef ZKHIP_AB_HOOKS
    if (const char* e = getenv("ZKHIP_LDE_SMALL")) { max_log = atoi(e); min_width = 1; }     // 0: off; else the largest log_n it takes, any width
#endif
    if (log_n > max_log || width < min_width) return ZKHIP_OK;
    const size_t n = (size_t)1 << log_n;
    const int B = 1 << log_blowup;
    LdeSmallArgs a{};
    a.in = in; a.in_ld = in_ld; a.out_ld = out_ld; a.ncols = width; a.cosets = (uint32_t)B;
    if (!lde_small_supported(a, log_n)) return ZKHIP_OK;
    const SmallPlan* sp;
    ZK_TRY(get_small_plan(ctx, log_n, &sp));
    a.tw_inv = sp->tw_inv; a.tw_fwd = sp->tw_fwd;
    const uint32_t wnb = two_adic_generator(log_n + log_blowup);
    for (int t = 0; t < B; t++) {
        ZK_TRY(get_small_pre(ctx, log_n, fmul(shift, fpow(wnb, (uint64_t)t)), &a.pre[t]));
        a.out[t] = out + (size_t)reverse_bits((uint32_t)t, log_blowup) * n * out_ld;
    }
    ZK_HIP(launch_lde_small(a, log_n, ctx->stream));
    *done = true;
    return ZKHIP_OK;
}

int op_coset_lde(zkhip_ctx* ctx, const uint32_t* in, size_t in_ld, uint32_t* out, size_t out_ld,
                 int log_n, uint32_t width, int log_blowup, uint32_t shift) {
    if (log_n < 0 || log_n > MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "coset_lde: log_n must be in [0, 22]");
    if (log_n > BIG_INNER_LOG) {
        if (log_blowup < 0 || log_blowup > 4 || log_n + log_blowup > TWO_ADICITY) return fail(ZKHIP_ERR_INVALID, "coset_lde: bad log_blowup");
        if (width == 0 || in_ld < width || out_ld < width) return fail(ZKHIP_ERR_INVALID, "coset_lde: bad width / ld");
        return coset_lde_big(ctx, in, in_ld, out, out_ld, log_n, width, log_blowup, shift);
    }
    if (log_blowup < 0 || log_blowup > 4 || log_n + log_blowup > TWO_ADICITY) return fail(ZKHIP_ERR_INVALID, "coset_lde: bad log_blowup");
    if (width == 0 || in_ld < width || out_ld < width) return fail(ZKHIP_ERR_INVALID, "coset_lde: bad width / ld");
    if (log_n < 5) {      // below the tile minimum: coefficients and evaluation by definition
        void* coef_s;
        ZK_TRY(ctx_reserve(ctx, S_COEF, ((size_t)1 << log_n) * width * 4, &coef_s));
        ZK_HIP(launch_small_eval(in, in_ld, (uint32_t*)coef_s, width, log_n, log_n, width, finv(two_adic_generator(log_n)), MONTY_R1,
                                 finv(to_monty(1u << log_n)), 0, ctx->stream));
        ZK_HIP(launch_small_eval((const uint32_t*)coef_s, width, out, out_ld, log_n, log_n + log_blowup, width,
                                 two_adic_generator(log_n + log_blowup), shift, MONTY_R1, 1, ctx->stream));
        return ZKHIP_OK;
    }
    {   // 2^11 .. 2^15 rows: one launch, any width
        bool done = false;
        ZK_TRY(lde_small(ctx, in, in_ld, out, out_ld, log_n, width, log_blowup, shift, &done));
        if (done) return ZKHIP_OK;
    }
    // A width that is not a multiple of 32 columns: the tile passes take two columns per lane (and, at 2^20 rows, the fused middle launch) only
    // for whole 32-column tiles -- a 612-column matrix went through the one-column-per-lane passes at 2.3 x the time of a 608-column one
    // (14.5 against 6.2 ms per 2^20-row LDE; round 5: the SHA-256 chip's width).  Columns are independent: the whole tiles go first, the
    // remainder follows as a narrow matrix of its own (the coefficient workspace is reused in stream order).
    if (width > 32 && width % 32 != 0) {
        const uint32_t w32 = width & ~31u;
        ZK_TRY(op_coset_lde(ctx, in, in_ld, out, out_ld, log_n, w32, log_blowup, shift));
        return op_coset_lde(ctx, in + w32, in_ld, out + w32, out_ld, log_n, width - w32, log_blowup, shift);
    }
    const size_t n = (size_t)1 << log_n;
    const int B = 1 << log_blowup;
    int m1, m2;
    split(log_n, &m1, &m2);
    // coefficients into the coefficient workspace
    void* coef_v;
    ZK_TRY(ctx_reserve(ctx, S_COEF, n * width * 4, &coef_v));
    uint32_t* coef = (uint32_t*)coef_v;
    const uint32_t wnb = two_adic_generator(log_n + log_blowup);
    if (m1 == 0) {
        ZK_TRY(run_inverse(ctx, in, in_ld, coef, width, log_n, width, /*transposed=*/false));
        for (int t = 0; t < B; t++) {
            const uint32_t st = fmul(shift, fpow(wnb, (uint64_t)t));
            uint32_t* dst = out + (size_t)reverse_bits((uint32_t)t, log_blowup) * n * out_ld;
            const NttPlan* p;
            ZK_TRY(get_plan(ctx, log_n, 1, st, &p));
            NttPassArgs a = base_args(ctx, coef, width, dst, out_ld, width, false);
            a.num_tiles = 1; a.log_m = log_n; a.in_tile_mul = 0; a.in_stride = 1; a.out_tile_mul = 0; a.out_stride = 1;
            a.pre = p->pre; a.bitrev_out = 1;
            ZK_HIP(launch_ntt_pass(a, false, ctx->stream));
        }
        return ZKHIP_OK;
    }
    uint32_t* dsts[16];
    uint32_t shifts[16];
    for (int t = 0; t < B; t++) {
        shifts[t] = fmul(shift, fpow(wnb, (uint64_t)t));
        dsts[t] = out + (size_t)reverse_bits((uint32_t)t, log_blowup) * n * out_ld;
    }
    return lde_two_pass(ctx, in, in_ld, coef, width, dsts, out_ld, log_n, width, shifts, B);
}

int op_merkle_commit(zkhip_ctx* ctx, const MatDesc* mats, int nmats, int log_h, uint32_t* tree) {
    if (nmats < 1 || nmats > MAX_LEAF_MATS) return fail(ZKHIP_ERR_INVALID, "merkle_commit: 1..8 matrices");
    if (log_h < 0 || log_h > 30) return fail(ZKHIP_ERR_INVALID, "merkle_commit: bad log_h");
    LeafArgs la{};
    for (int m = 0; m < nmats; m++) la.mats[m] = mats[m];
    la.nmats = nmats; la.height = (uint64_t)1 << log_h; la.digests = tree;
    uint32_t* level = tree;
    uint64_t count = la.height;
    if (count > COOP_TOP_NODES && count <= coop_max_nodes()) {
        // a medium tree: the workgroups that walk the subtrees hash their own leaves first (one launch less than leaves + subtrees + top)
        const uint32_t rest = count >= 4096 ? 128u : 32u;
        ZK_HIP(launch_hash_sub(la, (uint32_t)count / rest, ctx->stream));
        for (uint64_t c = count; c > rest; c >>= 1) level += 8 * c;
        ZK_HIP(launch_compress_top(level, rest, ctx->stream));
        return ZKHIP_OK;
    }
    ZK_HIP(launch_hash_rows(la, ctx->stream));
    while (count > COOP_TOP_NODES) {
        if (count <= coop_max_nodes()) {
            // medium levels: one launch reduces the level to `rest` nodes (each workgroup walks its own subtree), one more finishes.
            // 128 workgroups keep every level of a subtree within one sweep of a 1024-thread workgroup (64 permutations at a time)
            const uint32_t rest = count >= 4096 ? 128u : 32u;
            const uint32_t sub = (uint32_t)count / rest;
            ZK_HIP(launch_compress_sub(level, (uint32_t)count, sub, ctx->stream));
            for (uint64_t c = count; c > rest; c >>= 1) level += 8 * c;
            count = rest;
            break;
        }
        uint32_t* next = level + 8 * count;
        ZK_HIP(launch_compress_level(level, next, count / 2, ctx->stream));
        level = next; count >>= 1;
    }
    ZK_HIP(launch_compress_top(level, (uint32_t)count, ctx->stream));
    return ZKHIP_OK;
}

// p3-merkle-tree FieldMerkleTreeMmcs::commit with matrices of different (power-of-two) heights:
// the tallest ones form the leaves; a shorter matrix is injected at the level with as many
// nodes as it has rows: node = compress(node, sponge(row)).
int op_merkle_commit_mixed(zkhip_ctx* ctx, const MatDesc* mats, const int* log_heights, int nmats, uint32_t* tree) {
    if (nmats < 1 || nmats > 32) return fail(ZKHIP_ERR_INVALID, "merkle_commit_mixed: 1..32 matrices");
    int log_h = 0;
    for (int m = 0; m < nmats; m++) {
        if (log_heights[m] < 0 || log_heights[m] > 30) return fail(ZKHIP_ERR_INVALID, "merkle_commit_mixed: bad height");
        if (log_heights[m] > log_h) log_h = log_heights[m];
    }
    auto gather = [&](int lh, LeafArgs& la) {
        la = LeafArgs{};
        for (int m = 0; m < nmats; m++)
            if (log_heights[m] == lh) {
                if (la.nmats == MAX_LEAF_MATS) return false;
                la.mats[la.nmats++] = mats[m];
            }
        la.height = (uint64_t)1 << lh;
        return true;
    };
    LeafArgs la;
    if (!gather(log_h, la)) return fail(ZKHIP_ERR_INVALID, "merkle_commit_mixed: at most 8 matrices per height");
    if (la.nmats == nmats) return op_merkle_commit(ctx, la.mats, nmats, log_h, tree);      // one height: a plain commitment (and its fused launches)
    la.digests = tree;
    ZK_HIP(launch_hash_rows(la, ctx->stream));
    uint32_t* level = tree;
    for (int lvl = log_h - 1; lvl >= 0; lvl--) {
        const uint64_t cnt = (uint64_t)1 << lvl;
        // below the shortest matrix nothing is injected any more: the rest of the tree goes like a plain commitment (subtree + top launch)
        int min_lh = log_h;
        for (int m = 0; m < nmats; m++) if (log_heights[m] < min_lh) min_lh = log_heights[m];
        if (lvl < min_lh && 2 * cnt <= coop_max_nodes()) {
            uint64_t count = 2 * cnt;                        // nodes of the current level
            if (count > COOP_TOP_NODES) {
                const uint32_t rest = count >= 4096 ? 128u : 32u;
                ZK_HIP(launch_compress_sub(level, (uint32_t)count, (uint32_t)count / rest, ctx->stream));
                for (uint64_t c = count; c > rest; c >>= 1) level += 8 * c;
                count = rest;
            }
            ZK_HIP(launch_compress_top(level, (uint32_t)count, ctx->stream));
            return ZKHIP_OK;
        }
        uint32_t* next = level + 16 * cnt;
        LeafArgs inj;
        if (!gather(lvl, inj)) return fail(ZKHIP_ERR_INVALID, "merkle_commit_mixed: at most 8 matrices per height");
        if (inj.nmats && compress_inject_ok(inj)) {           // parents, row digests and the injection in one launch
            inj.digests = next;
            ZK_HIP(launch_compress_inject(level, inj, ctx->stream));
            level = next;
            continue;
        }
        ZK_HIP(launch_compress_level(level, next, cnt, ctx->stream));
        if (inj.nmats) {
            void* tmp;
            ZK_TRY(ctx_reserve(ctx, S_TMP, cnt * 32, &tmp));
            inj.digests = (uint32_t*)tmp;
            ZK_HIP(launch_hash_rows(inj, ctx->stream));
            ZK_HIP(launch_inject(next, (const uint32_t*)tmp, cnt, ctx->stream));
        }
        level = next;
    }
    return ZKHIP_OK;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zkhip_version(void) { return ZKHIP_VERSION; }
const char* zkhip_last_error(void) { return g_last_error.c_str(); }

// ---- LOGICAL devices (A/B build only, never in libzkhip.so).  The multi-device entries (jobs.cpp) deal shards over a device list, keep
// one context pool per listed device and expect a shard's trace to live on the device the shard is dealt to.  On a one-GPU box that
// code path cannot run (the list refuses a repeated ordinal).  With ZKHIP_LOGICAL_DEVICES=K in the environment of a process that
// loaded libzkhip_ab.so, ordinals 0 .. K-1 are K LOGICAL devices on the physical ones (ordinal d -> physical d mod visible): own
// pools, own workers and lanes, and allocations made through zkhip_malloc remember the logical device of their context, so that
// "the trace lives where the shard is dealt" is checked by logical ordinal (prove_shards_on).
extern "C++" {
#ifdef ZKHIP_AB_HOOKS
namespace zk {
int logical_devices() {
    static const int k = [] { const char* e = getenv("ZKHIP_LOGICAL_DEVICES"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 64 ? v : 0; }();
    return k;
}
static std::mutex g_alloc_mu;
static std::vector<std::pair<std::pair<uintptr_t, size_t>, int>> g_allocs;     // ((base, bytes), logical device)
int logical_device_of(const void* p) {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    const uintptr_t a = (uintptr_t)p;
    for (auto& e : g_allocs) if (a >= e.first.first && a < e.first.first + e.first.second) return e.second;
    return -1;
}
}  // namespace zk
#endif
namespace zk {
int physical_device(int device) {
#ifdef ZKHIP_AB_HOOKS
    if (logical_devices() > 0 && device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return device; }
        return device % n;
    }
#endif
    return device;
}
}  // namespace zk
}  // extern "C++"

int zkhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
#ifdef ZKHIP_AB_HOOKS
    if (n > 0 && logical_devices() > 0) return logical_devices();
#endif
    return n;
}

int zkhip_ctx_create(int device, void* stream, zkhip_ctx** out) {
    if (!out) return fail(ZKHIP_ERR_INVALID, "ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(ZKHIP_ERR_NO_DEVICE, "no HIP device visible: libzkhip has no CPU fallback");
    }
    const int logical = device;
#ifdef ZKHIP_AB_HOOKS
    if (logical_devices() > 0) {
        if (device < 0 || device >= logical_devices()) return fail(ZKHIP_ERR_INVALID, "ctx_create: logical device ordinal out of range");
        device = physical_device(device);
    }
#endif
    if (device < 0 || device >= n) return fail(ZKHIP_ERR_INVALID, "ctx_create: device ordinal out of range");
    ZK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ZK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ZKHIP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libzkhip is built for gfx950 only");
    zkhip_ctx* ctx = new (std::nothrow) zkhip_ctx();
    if (!ctx) return fail(ZKHIP_ERR_NOMEM, "ctx_create: out of host memory");
    g_live_contexts.fetch_add(1);
    ctx->device = device;
    ctx->logical_device = logical;
    if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; g_live_contexts.fetch_sub(1); return hip_fail(e, "hipStreamCreate"); }
        ctx->own_stream = true;
    }
    int rc = ZKHIP_OK;
    do {
        hipError_t e;
        if ((e = hipMalloc((void**)&ctx->w1024_fwd, 1024 * 4)) != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        if ((e = hipMalloc((void**)&ctx->w1024_inv, 1024 * 4)) != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        const uint32_t w = two_adic_generator(10);
        if ((e = launch_pow_table(ctx->w1024_fwd, 1024, w, MONTY_R1, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "pow_table"); break; }
        if ((e = launch_pow_table(ctx->w1024_inv, 1024, finv(w), MONTY_R1, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "pow_table"); break; }
        if ((e = hipMalloc((void**)&ctx->w1024f_fwd, 1024 * 4)) != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        if ((e = hipMalloc((void**)&ctx->w1024f_inv, 1024 * 4)) != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        if ((e = launch_fused_table(ctx->w1024f_fwd, ctx->w1024_fwd, 1, 0, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "fused_table"); break; }
        if ((e = launch_fused_table(ctx->w1024f_inv, ctx->w1024_inv, 1, 0, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "fused_table"); break; }
        // the Poseidon2 tables in effect (built-in or zkhip_load_poseidon2_params) go to this device's constant memory
        if ((e = hash_upload_p2_tables(g_p2_tables, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "upload Poseidon2 tables"); break; }
        if ((e = stark_upload_p2_tables(g_p2_tables, ctx->stream)) != hipSuccess) { rc = hip_fail(e, "upload Poseidon2 tables"); break; }
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { rc = hip_fail(e, "sync"); break; }
    } while (0);
    if (rc != ZKHIP_OK) { zkhip_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return ZKHIP_OK;
}

void zkhip_ctx_destroy(zkhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->sha_key) zkhip_machine_key_destroy(ctx->sha_key);
    if (ctx->rec_key) zkhip_machine_key_destroy(ctx->rec_key);
    if (ctx->fri_graph_exec) (void)hipGraphExecDestroy(ctx->fri_graph_exec);
    for (NttPlan& p : ctx->plans) { if (p.pre) (void)hipFree(p.pre); if (p.post) (void)hipFree(p.post); if (p.pre_f) (void)hipFree(p.pre_f); if (p.post_f) (void)hipFree(p.post_f); }
    if (ctx->w1024f_fwd) (void)hipFree(ctx->w1024f_fwd);
    if (ctx->w1024f_inv) (void)hipFree(ctx->w1024f_inv);
    for (BigPlan& p : ctx->big_plans) if (p.tw) (void)hipFree(p.tw);
    for (SmallPlan& p : ctx->small_plans) { if (p.tw_inv) (void)hipFree(p.tw_inv); if (p.tw_fwd) (void)hipFree(p.tw_fwd); }
    for (SmallPre& p : ctx->small_pres) if (p.pre) (void)hipFree(p.pre);
    for (ColPlan& p : ctx->col_plans) { if (p.pre) (void)hipFree(p.pre); if (p.post2d) (void)hipFree(p.post2d); }
    for (DeviceBuffer& b : ctx->scratch) if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->w1024_fwd) (void)hipFree(ctx->w1024_fwd);
    if (ctx->w1024_inv) (void)hipFree(ctx->w1024_inv);
    for (auto& d : ctx->domains) { (void)hipFree(d.xs); (void)hipFree(d.sel_first); (void)hipFree(d.sel_last); (void)hipFree(d.itw); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    g_live_contexts.fetch_sub(1);
}

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

int zkhip_ctx_set_lde_fusion(zkhip_ctx* ctx, int on) {
    CHECK_CTX(ctx);
    const int prev = ctx->lde_fusion ? 1 : 0;
    ctx->lde_fusion = on != 0;
    return prev;
}
int zkhip_ctx_sync(zkhip_ctx* ctx) { CHECK_CTX(ctx); return dev_sync(ctx); }
// hipDeviceScheduleBlockingSync is the one switch this runtime honours (measured, tools/waitprobe: with it a waiting thread uses
// 0.001 s of CPU per 0.2 s waited, without it 0.2 s -- hipStreamSynchronize, hipEventSynchronize and the wait inside a copy to
// pageable memory alike; the per-event flag hipEventBlockingSync changes nothing).  It must be set before the device is first used.
int zkhip_set_wait_mode(int blocking, int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(ZKHIP_ERR_NO_DEVICE, "set_wait_mode: no HIP device visible"); }
    if (device >= n) return fail(ZKHIP_ERR_INVALID, "set_wait_mode: no such device");
    if (g_live_contexts.load() != 0) return fail(ZKHIP_ERR_INVALID, "set_wait_mode: contexts exist (the mode is fixed when a device is first used)");
    int cur = 0;
    (void)hipGetDevice(&cur);
    int rc = ZKHIP_OK;
    for (int d = device < 0 ? 0 : device; d < (device < 0 ? n : device + 1); d++) {      // (a rank of eight should not touch the other seven devices)
        hipError_t e = hipSetDevice(d);
        if (e == hipSuccess) e = hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = hip_fail(e, "hipSetDeviceFlags (too late: the device is already in use?)"); }
    }
    (void)hipSetDevice(device < 0 ? cur : device);
    return rc;
}
void* zkhip_ctx_stream(zkhip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int zkhip_malloc(zkhip_ctx* ctx, size_t bytes, void** d_ptr) {
    CHECK_CTX(ctx);
    if (!d_ptr) return fail(ZKHIP_ERR_INVALID, "malloc: null out pointer");
    ZK_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
#ifdef ZKHIP_AB_HOOKS
    if (logical_devices() > 0) { std::lock_guard<std::mutex> lk(g_alloc_mu); g_allocs.push_back({{(uintptr_t)*d_ptr, bytes ? bytes : 4}, ctx->logical_device}); }
#endif
    return ZKHIP_OK;
}
int zkhip_free(zkhip_ctx* ctx, void* d_ptr) {
    CHECK_CTX(ctx);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
#ifdef ZKHIP_AB_HOOKS
    if (logical_devices() > 0 && d_ptr) {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        for (size_t i = 0; i < g_allocs.size(); i++) if (g_allocs[i].first.first == (uintptr_t)d_ptr) { g_allocs.erase(g_allocs.begin() + (long)i); break; }
    }
#endif
    if (d_ptr) ZK_HIP(hipFree(d_ptr));
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d(zkhip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    CHECK_CTX(ctx);
    ZK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_memcpy_d2h(zkhip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    CHECK_CTX(ctx);
    ZK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_to_monty(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n) {
    CHECK_CTX(ctx);
    ZK_HIP(launch_convert(d_in, d_out, n, true, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_from_monty(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n) {
    CHECK_CTX(ctx);
    ZK_HIP(launch_convert(d_in, d_out, n, false, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_fill_uniform(zkhip_ctx* ctx, uint64_t seed, int log_n, uint32_t width, uint32_t* d_out, size_t ld) {
    CHECK_CTX(ctx);
    if (log_n < 0 || log_n > 30 || width == 0 || ld < width || !d_out) return fail(ZKHIP_ERR_INVALID, "fill_uniform: bad arguments");
    ZK_HIP(launch_fill_uniform(d_out, ld, seed, (uint64_t)1 << log_n, width, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_gen_trace(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, int log_n, uint32_t width, uint32_t* d_out, size_t ld) {
    CHECK_CTX(ctx);
    if (log_n < 0 || log_n > 30 || width == 0 || width % 4 != 0 || ld < width || !d_out)
        return fail(ZKHIP_ERR_INVALID, "gen_trace: width must be a positive multiple of 4");
    ZK_HIP(launch_gen_trace(d_out, ld, seed + shard, (uint64_t)1 << log_n, width, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_gen_trace_logup(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, int log_n, uint32_t width, int pairs, uint32_t* d_out, size_t ld) {
    CHECK_CTX(ctx);
    if (log_n < 0 || log_n > 30 || width == 0 || width % 4 != 0 || ld < width || !d_out || pairs < 0 || (uint32_t)pairs * 8 > width)
        return fail(ZKHIP_ERR_INVALID, "gen_trace_logup: width must be a multiple of 4 and hold 2 groups per pair");
    ZK_HIP(launch_gen_trace_logup(d_out, ld, seed + shard, (uint64_t)1 << log_n, width, (uint32_t)pairs, seed + shard, width, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_gen_trace_logup_cross(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, uint64_t partner_shard, int log_n, uint32_t width,
                                uint32_t partner_width, int pairs, uint32_t* d_out, size_t ld) {
    CHECK_CTX(ctx);
    if (log_n < 0 || log_n > 30 || width == 0 || width % 4 != 0 || ld < width || !d_out || pairs < 1 || (uint32_t)pairs * 8 > width ||
        partner_width % 4 != 0 || (uint32_t)pairs * 8 > partner_width)
        return fail(ZKHIP_ERR_INVALID, "gen_trace_logup_cross: both tables need 2 groups per pair");
    ZK_HIP(launch_gen_trace_logup(d_out, ld, seed + shard, (uint64_t)1 << log_n, width, (uint32_t)pairs, seed + partner_shard, partner_width, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_dft(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_ld, uint32_t* d_out, size_t out_ld,
              int log_n, uint32_t width, int inverse, int bitrev_out) {
    CHECK_CTX(ctx);
    if (log_n < 0 || log_n > MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "dft: log_n must be in [0, 22]");
    if (!d_in || !d_out || width == 0 || in_ld < width || out_ld < width) return fail(ZKHIP_ERR_INVALID, "dft: bad arguments");
    if (log_n < 5) {      // below the tile minimum: by definition (must not run in place)
        if (d_in == d_out) return fail(ZKHIP_ERR_INVALID, "dft: transforms of fewer than 32 rows are out of place");
        if (inverse && bitrev_out) return fail(ZKHIP_ERR_INVALID, "dft: inverse transform writes natural order only");
        const uint32_t w = inverse ? finv(two_adic_generator(log_n)) : two_adic_generator(log_n);
        const uint32_t scale = inverse ? finv(to_monty(1u << log_n)) : MONTY_R1;
        ZK_HIP(launch_small_eval(d_in, in_ld, d_out, out_ld, log_n, log_n, width, w, MONTY_R1, scale, bitrev_out, ctx->stream));
        return ZKHIP_OK;
    }
    if (inverse) {
        if (bitrev_out) return fail(ZKHIP_ERR_INVALID, "dft: inverse transform writes natural order only");
        if (log_n > BIG_INNER_LOG) return run_inverse_big(ctx, d_in, in_ld, d_out, out_ld, log_n, width);
        return run_inverse(ctx, d_in, in_ld, d_out, out_ld, log_n, width, false);
    }
    if (log_n > BIG_INNER_LOG) return run_forward_natural_big(ctx, d_in, in_ld, d_out, out_ld, log_n, width, MONTY_R1, bitrev_out != 0);
    return run_forward_natural(ctx, d_in, in_ld, d_out, out_ld, log_n, width, MONTY_R1, bitrev_out != 0);
}

int zkhip_coset_lde(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_ld, uint32_t* d_out, size_t out_ld,
                    int log_n, uint32_t width, int log_blowup, uint32_t shift) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out) return fail(ZKHIP_ERR_INVALID, "coset_lde: null pointer");
    if (shift == 0 || shift >= P) return fail(ZKHIP_ERR_INVALID, "coset_lde: shift must be a canonical non-zero element");
    return op_coset_lde(ctx, d_in, in_ld, d_out, out_ld, log_n, width, log_blowup, to_monty(shift));
}

int zkhip_ntt_pass(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t ld, int log_n, uint32_t width, int which) {
    CHECK_CTX(ctx);
    if (log_n < 11 || log_n > 20) return fail(ZKHIP_ERR_INVALID, "ntt_pass: log_n must be in [11, 20]");
    if (which >= 2 && which <= 5) {
        // one launch of the trace LDE exactly as zkhip_prove_shard enqueues it, on THIS context's workspaces (the buffers its
        // proofs use: same sizes, so nothing is reallocated or moved) -- the in-proof placement of the roofline kernel
        if (width == 0 || ld < width || (which == 2 && !d_in)) return fail(ZKHIP_ERR_INVALID, "ntt_pass: bad arguments");
        const size_t n = (size_t)1 << log_n;
        void *coef, *lde;
        ZK_TRY(ctx_reserve(ctx, S_COEF, n * width * 4, &coef));
        ZK_TRY(ctx_reserve(ctx, S_TLDE, 2 * n * width * 4, &lde));
        NttPassArgs a;
        bool inv;
        ZK_TRY(lde_pass_args(ctx, which - 2, d_in, ld, (uint32_t*)coef, width, (uint32_t*)lde, width, log_n, width, MONTY_GEN, &a, &inv));
        a.bench_tag = 1;
        ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
        return ZKHIP_OK;
    }
    if (which == 6 || which == 7) {
        // the block-form first inverse pass and the fused middle launch, on the same workspaces
        if (width == 0 || ld < width || (which == 6 && !d_in)) return fail(ZKHIP_ERR_INVALID, "ntt_pass: bad arguments");
        const size_t n = (size_t)1 << log_n;
        if (!lde_fused_shape(log_n, width, width, width, 2)) return fail(ZKHIP_ERR_INVALID, "ntt_pass: which = 6, 7 take 2^20 rows x a multiple of 32 columns");
        void *coef, *lde;
        ZK_TRY(ctx_reserve(ctx, S_COEF, n * width * 4, &coef));
        ZK_TRY(ctx_reserve(ctx, S_TLDE, 2 * n * width * 4, &lde));
        if (which == 6) {
            NttPassArgs a;
            bool inv;
            ZK_TRY(lde_pass_args(ctx, LDE_I1_BLOCKS, d_in, ld, (uint32_t*)coef, width, nullptr, 0, log_n, width, 0, &a, &inv));
            a.bench_tag = 1;
            ZK_HIP(launch_ntt_pass(a, inv, ctx->stream));
            return ZKHIP_OK;
        }
        const uint32_t wnb = two_adic_generator(log_n + 1);
        const uint32_t shifts[2] = {MONTY_GEN, fmul(MONTY_GEN, wnb)};
        uint32_t* dsts[2] = {(uint32_t*)lde, (uint32_t*)lde + n * width};
        LdeFusedArgs f;
        const int frc = lde_fused_args(ctx, (const uint32_t*)coef, width, dsts, width, log_n, width, shifts, &f);
        if (frc != ZKHIP_OK) return frc == ZKHIP_ERR_UNSUPPORTED_FUSION ? ZKHIP_ERR_INVALID : frc;      // (the private code never crosses the ABI)
        f.bench_tag = 1;
        ZK_HIP(launch_lde_fused(f, ctx->stream));
        return ZKHIP_OK;
    }
    if (which != 0 && which != 1) return fail(ZKHIP_ERR_INVALID, "ntt_pass: which must be 0 .. 7");
    if (!d_in || !d_out || width == 0 || ld < width) return fail(ZKHIP_ERR_INVALID, "ntt_pass: bad arguments");
    const NttPlan* p;
    ZK_TRY(get_plan(ctx, log_n, 1, MONTY_R1, &p));
    const uint64_t M1 = 1ull << p->m1, M2 = 1ull << p->m2;
    NttPassArgs a = base_args(ctx, d_in, ld, d_out, ld, width, false);
    if (which == 0) {
        a.num_tiles = (uint32_t)M1; a.log_m = p->m2;
        a.in_tile_mul = 1; a.in_stride = M1; a.out_tile_mul = 1; a.out_stride = M1; a.post = p->post; a.bitrev_out = 1;
    } else {
        a.num_tiles = (uint32_t)M2; a.log_m = p->m1;
        a.in_tile_mul = M1; a.in_stride = 1; a.out_tile_mul = M1; a.out_stride = 1; a.bitrev_out = 1;
    }
    ZK_HIP(launch_ntt_pass(a, false, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_poseidon2_permute(zkhip_ctx* ctx, uint32_t* d_states, size_t count) {
    CHECK_CTX(ctx);
    if (!d_states && count) return fail(ZKHIP_ERR_INVALID, "poseidon2_permute: null pointer");
    ZK_HIP(launch_permute_states(d_states, count, ctx->stream));
    return ZKHIP_OK;
}

static int make_descs(const uint32_t* const* d_mats, const size_t* lds, const uint32_t* widths, int nmats, MatDesc* out) {
    if (nmats < 1 || nmats > MAX_LEAF_MATS || !d_mats || !lds || !widths) return fail(ZKHIP_ERR_INVALID, "1..8 matrices required");
    for (int m = 0; m < nmats; m++) {
        if (!d_mats[m] || lds[m] < widths[m]) return fail(ZKHIP_ERR_INVALID, "bad matrix descriptor");
        out[m].ptr = d_mats[m]; out[m].ld = lds[m]; out[m].width = widths[m];
    }
    return ZKHIP_OK;
}

int zkhip_hash_rows(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds, const uint32_t* widths,
                    int nmats, size_t height, uint32_t* d_digests) {
    CHECK_CTX(ctx);
    LeafArgs la{};
    ZK_TRY(make_descs(d_mats, lds, widths, nmats, la.mats));
    if (!d_digests) return fail(ZKHIP_ERR_INVALID, "hash_rows: null output");
    la.nmats = nmats; la.height = height; la.digests = d_digests;
    ZK_HIP(launch_hash_rows(la, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_merkle_commit(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds, const uint32_t* widths,
                        int nmats, int log_h, uint32_t* d_tree) {
    CHECK_CTX(ctx);
    MatDesc descs[MAX_LEAF_MATS];
    ZK_TRY(make_descs(d_mats, lds, widths, nmats, descs));
    if (!d_tree) return fail(ZKHIP_ERR_INVALID, "merkle_commit: null output");
    return op_merkle_commit(ctx, descs, nmats, log_h, d_tree);
}

// ---- RISC Zero Hal layout (column-major [count][size]).  2^20-point polynomials (the segment size of the reference's RISC Zero
// backend: po2 = 20, benchmark.md:36) run NATIVELY on the contiguous vectors: two launches of ntt_colpass_kernel per transform, the
// four-step transpose folded into the tile's load / store (ntt.hip) -- 16 B per element and transform, no transpose pass.  Other
// sizes keep the transposing adapter (the polynomials become the columns of a row-major matrix: 32 B per element).
static int get_col_plan(zkhip_ctx* ctx, int inverse, uint32_t shift, const ColPlan** out) {
    for (const ColPlan& p : ctx->col_plans)
        if (p.inverse == inverse && (inverse || p.shift == shift)) { *out = &p; return ZKHIP_OK; }
    ColPlan p;
    p.inverse = inverse; p.shift = shift;
    const uint32_t w = two_adic_generator(20);
    ZK_HIP(hipMalloc((void**)&p.post2d, (size_t)4 << 20));
    if (inverse) {
        // first inverse pass (over kc -> c, columns kr): times w^-(c kr) / N
        ZK_HIP(launch_post2d_table(p.post2d, finv(w), MONTY_R1, finv(to_monty(1u << 20)), ctx->stream));
    } else {
        // first forward pass (over r -> kr, columns c): input row r times (s^1024)^r, output (kr, c) times w^(c kr) s^c
        ZK_HIP(launch_post2d_table(p.post2d, w, shift, MONTY_R1, ctx->stream));
        if (shift != MONTY_R1) {
            ZK_HIP(hipMalloc((void**)&p.pre, 1024 * 4));
            ZK_HIP(launch_pow_table(p.pre, 1024, fpow(shift, 1024), MONTY_R1, ctx->stream));
        }
    }
    ctx->col_plans.push_back(p);
    *out = &ctx->col_plans.back();
    return ZKHIP_OK;
}

int zkhip_batch_interpolate_colmajor(zkhip_ctx* ctx, const uint32_t* d_evals, uint32_t* d_coeffs, uint32_t count, int log_size) {
    CHECK_CTX(ctx);
    if (!d_evals || !d_coeffs || count == 0 || log_size < 5 || log_size > 20) return fail(ZKHIP_ERR_INVALID, "batch_interpolate_colmajor: bad arguments");
    const uint64_t n = (uint64_t)1 << log_size;
    void *a, *b;
    ZK_TRY(ctx_reserve(ctx, S_COL_A, n * count * 4, &a));
    if (log_size == 20 && (reinterpret_cast<uintptr_t>(d_evals) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_coeffs) & 7) == 0) {
        // x[c + 1024 r] = 1/N sum_kr w_R^-(r kr) [ w_N^-(c kr) sum_kc w_C^-(c kc) X[kr + 1024 kc] ],  X[k] stored at bitrev(k)
        const ColPlan* p;
        ZK_TRY(get_col_plan(ctx, 1, 0, &p));
        ColPassArgs q{};
        q.in = d_evals; q.out = (uint32_t*)a; q.in_batch = q.out_batch = n; q.count = count;
        q.tload = 1; q.in_brev = 1; q.w1024 = ctx->w1024_inv; q.post2d = p->post2d;
        ZK_HIP(launch_ntt_colpass(q, true, ctx->stream));
        ColPassArgs r{};
        r.in = (const uint32_t*)a; r.out = d_coeffs; r.in_batch = r.out_batch = n; r.count = count;
        r.tload = 1; r.w1024 = ctx->w1024_inv;
        ZK_HIP(launch_ntt_colpass(r, true, ctx->stream));
        return ZKHIP_OK;
    }
    ZK_TRY(ctx_reserve(ctx, S_COL_B, n * count * 4, &b));
    // evaluations arrive bit-reversed (Hal convention): undo it while transposing
    ZK_HIP(launch_transpose(d_evals, (uint32_t*)a, count, n, log_size, 0, ctx->stream));
    ZK_TRY(run_inverse(ctx, (const uint32_t*)a, count, (uint32_t*)b, count, log_size, count, false));
    ZK_HIP(launch_transpose((const uint32_t*)b, d_coeffs, n, count, 0, 0, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_batch_expand_colmajor(zkhip_ctx* ctx, const uint32_t* d_coeffs, uint32_t* d_evals, uint32_t count, int log_size,
                                int log_blowup, uint32_t shift) {
    CHECK_CTX(ctx);
    if (!d_coeffs || !d_evals || count == 0 || log_size < 5 || log_size > 20 || log_blowup < 0 || log_blowup > 4 ||
        log_size + log_blowup > 24 || shift == 0 || shift >= P)
        return fail(ZKHIP_ERR_INVALID, "batch_expand_colmajor: bad arguments");
    const uint64_t n = (uint64_t)1 << log_size, m = n << log_blowup;
    const uint32_t sm = to_monty(shift), wnb = two_adic_generator(log_size + log_blowup);
    void *a, *b;
    ZK_TRY(ctx_reserve(ctx, S_COL_A, n * count * 4, &a));
    if (log_size == 20 && (reinterpret_cast<uintptr_t>(d_evals) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_coeffs) & 7) == 0) {
        // a zero-padded size-m transform = 2^b coset transforms of size n; coset t fills the block bitrev_b(t) of the bit-reversed output
        for (int t = 0; t < (1 << log_blowup); t++) {
            const uint32_t st = fmul(sm, fpow(wnb, (uint64_t)t));
            const ColPlan* p;
            ZK_TRY(get_col_plan(ctx, 0, st, &p));
            ColPassArgs q{};
            q.in = d_coeffs; q.out = (uint32_t*)a; q.in_batch = q.out_batch = n; q.count = count;
            q.tstore = 1; q.w1024 = ctx->w1024_fwd; q.pre = p->pre; q.post2d = p->post2d;
            ZK_HIP(launch_ntt_colpass(q, false, ctx->stream));
            ColPassArgs r{};
            r.in = (const uint32_t*)a; r.out = d_evals + (size_t)reverse_bits((uint32_t)t, log_blowup) * n;
            r.in_batch = n; r.out_batch = m; r.count = count;
            r.tstore = 1; r.out_brev = 1; r.w1024 = ctx->w1024_fwd;
            ZK_HIP(launch_ntt_colpass(r, false, ctx->stream));
        }
        return ZKHIP_OK;
    }
    ZK_TRY(ctx_reserve(ctx, S_COL_B, m * count * 4, &b));
    ZK_HIP(launch_transpose(d_coeffs, (uint32_t*)a, count, n, 0, 0, ctx->stream));
    // zero-padded size-m transform = 2^b coset transforms of size n (as in op_coset_lde)
    for (int t = 0; t < (1 << log_blowup); t++) {
        const uint32_t st = fmul(sm, fpow(wnb, (uint64_t)t));
        uint32_t* dst = (uint32_t*)b + (size_t)reverse_bits((uint32_t)t, log_blowup) * n * count;
        ZK_TRY(run_forward_natural(ctx, (const uint32_t*)a, count, dst, count, log_size, count, st, true));
    }
    ZK_HIP(launch_transpose((const uint32_t*)b, d_evals, m, count, 0, 0, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_merkle_commit_p24_colmajor(zkhip_ctx* ctx, const uint32_t* d_mat, uint32_t cols, int log_rows, uint32_t* d_tree) {
    CHECK_CTX(ctx);
    if (!d_mat || !d_tree || cols == 0 || log_rows < 0 || log_rows > 28) return fail(ZKHIP_ERR_INVALID, "merkle_commit_p24_colmajor: bad arguments");
    ZK_HIP(launch_merkle_p24_colmajor(d_mat, cols, log_rows, d_tree, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_merkle_commit_mixed(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds, const uint32_t* widths,
                              const int* log_heights, int nmats, uint32_t* d_tree) {
    CHECK_CTX(ctx);
    if (nmats < 1 || nmats > 16 || !d_mats || !lds || !widths || !log_heights || !d_tree)
        return fail(ZKHIP_ERR_INVALID, "merkle_commit_mixed: bad arguments");
    MatDesc descs[16];
    for (int m = 0; m < nmats; m++) {
        if (!d_mats[m] || lds[m] < widths[m]) return fail(ZKHIP_ERR_INVALID, "bad matrix descriptor");
        descs[m].ptr = d_mats[m]; descs[m].ld = lds[m]; descs[m].width = widths[m];
    }
    return op_merkle_commit_mixed(ctx, descs, log_heights, nmats, d_tree);
}

}  // extern "C"
