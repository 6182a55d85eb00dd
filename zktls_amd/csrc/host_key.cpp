// host_key.cpp -- the commitment to a machine's PREPROCESSED columns computed on the HOST: no context, no device, no HIP call.
//
// The reference checks a proof on the CPU (client.verify, crates/guest-prover-sp1/src/sp1.rs:120) against a verifying key that
// `client.setup` (sp1.rs:113) derives from the program alone.  Here the key of a keyed machine is the Merkle root over the low-degree
// extensions of its preprocessed traces (zkhip_machine_setup, prover.cpp); until round 5 only a GPU context could compute it, so a party
// without an MI355X could check a compressed proof only against a key somebody else handed over.  This file restates the two steps
// -- coset LDE (p3-dft Radix2Dit coset_lde_batch + bit_reverse_rows, Cargo.lock:3903) and the mixed-height Poseidon2 commitment
// (p3-merkle-tree FieldMerkleTreeMmcs, Cargo.lock:4013) -- in plain host code: one column at a time through a radix-2 transform, sixteen
// sponges / compressions per AVX-512 register (p2_x16.cpp; scalar otherwise), a few threads.  Same words as the device path, bit for bit
// (tests/test_recursion_cpu.py against the oracle's key, tests/test_gpu_recursion.py against the device's).
#include <algorithm>
#include <new>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"
#include "../../include/zkhip_hal.h"
#include "babybear.cuh"
#include "context.h"
#include "p2_x16.h"
#include "poseidon2.cuh"

namespace zk {
namespace {

unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 4;
    return n > 16 ? 16u : n;          // (the boxes of this pool give a container 16 cores of the 256 they show)
}
template <class F>
void parallel_for(size_t n, size_t grain, const F& f) {
    const size_t chunks = (n + grain - 1) / grain;
    const unsigned nt = (unsigned)std::min<size_t>(host_threads(), chunks);
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> ts;
    const size_t per = ((chunks + nt - 1) / nt) * grain;
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
        if (lo < hi) ts.emplace_back([&f, lo, hi] { f(lo, hi); });
    }
    for (auto& t : ts) t.join();
}

// in-place radix-2 decimation-in-time transform of one contiguous column, natural order in and out; tw[j] = w^j, j < n / 2
void ntt_column(uint32_t* a, int log_n, const uint32_t* tw) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        const size_t j = reverse_bits((uint32_t)i, log_n);
        if (i < j) std::swap(a[i], a[j]);
    }
    for (int layer = 0; layer < log_n; layer++) {
        const size_t half = (size_t)1 << layer, step = n / (2 * half);
        for (size_t g = 0; g < n; g += 2 * half)
            for (size_t j = 0; j < half; j++) {
                const uint32_t t = fmul(a[g + half + j], tw[j * step]), u = a[g + j];
                a[g + j] = fadd(u, t);
                a[g + half + j] = fsub(u, t);
            }
    }
}
std::vector<uint32_t> twiddles(int log_n, bool inverse) {
    const size_t n = (size_t)1 << log_n;
    std::vector<uint32_t> tw(n > 1 ? n / 2 : 1);
    uint32_t root = two_adic_generator(log_n);
    if (inverse) root = finv(root);
    tw[0] = MONTY_R1;
    for (size_t i = 1; i < tw.size(); i++) tw[i] = fmul(tw[i - 1], root);
    return tw;
}

}  // namespace

// out[bitrev(i)][c] = f_c(shift w_m^i), i < m = n 2^b, where f_c interpolates column c of `in` (rows natural) over the n-th roots of unity.
// in: [n][ld] Montgomery words; out: [m][w]; shift in Montgomery form.
void host_coset_lde(const uint32_t* in, size_t ld, int log_n, uint32_t w, int b, uint32_t shift, uint32_t* out) {
    const size_t n = (size_t)1 << log_n, m = n << b;
    const int log_m = log_n + b;
    const std::vector<uint32_t> itw = twiddles(log_n, true), ftw = twiddles(log_m, false);
    std::vector<uint32_t> spow(n);                         // shift^j / n
    spow[0] = finv(to_monty((uint32_t)(n % P)));
    for (size_t j = 1; j < n; j++) spow[j] = fmul(spow[j - 1], shift);
    parallel_for(w, 1, [&](size_t lo, size_t hi) {
        std::vector<uint32_t> col(m);
        for (size_t c = lo; c < hi; c++) {
            for (size_t i = 0; i < n; i++) col[i] = in[i * ld + c];
            ntt_column(col.data(), log_n, itw.data());
            for (size_t j = 0; j < n; j++) col[j] = fmul(col[j], spow[j]);
            std::fill(col.begin() + (long)n, col.end(), 0u);
            ntt_column(col.data(), log_m, ftw.data());
            for (size_t i = 0; i < m; i++) out[(size_t)reverse_bits((uint32_t)i, log_m) * w + c] = col[i];
        }
    });
}

struct HostMat { const uint32_t* ptr; size_t ld; uint32_t width; int log_h; };

namespace {
// digests[r] = sponge over the concatenated row r of the matrices (overwrite mode, rate 8), r in [lo, hi), 16 rows per AVX-512 pass
void sponge_rows(const HostMat* mats, int nm, size_t lo, size_t hi, uint32_t* digests) {
    uint32_t total = 0;
    for (int k = 0; k < nm; k++) total += mats[k].width;
    const bool wide = p2x16_available();
    std::vector<uint32_t> rowbuf((size_t)total * 16);
    for (size_t r0 = lo; r0 < hi; r0 += 16) {
        const size_t cnt = std::min<size_t>(16, hi - r0);
        for (size_t l = 0; l < cnt; l++) {
            uint32_t* d = rowbuf.data() + l * total;
            for (int k = 0; k < nm; k++) { std::memcpy(d, mats[k].ptr + (r0 + l) * mats[k].ld, (size_t)mats[k].width * 4); d += mats[k].width; }
        }
        if (wide) {
            alignas(64) uint32_t st[16][16];
            std::memset(st, 0, sizeof st);
            for (uint32_t q = 0; q < total; q += 8) {
                for (uint32_t i = 0; i < 8 && q + i < total; i++)
                    for (size_t l = 0; l < cnt; l++) st[i][l] = rowbuf[l * total + q + i];
                p2x16_permute(st);
            }
            for (size_t l = 0; l < cnt; l++) for (int i = 0; i < 8; i++) digests[(r0 + l) * 8 + i] = st[i][l];
        } else {
            for (size_t l = 0; l < cnt; l++) {
                uint32_t s[16] = {};
                for (uint32_t q = 0; q < total; q += 8) {
                    for (uint32_t i = 0; i < 8 && q + i < total; i++) s[i] = rowbuf[l * total + q + i];
                    p2_permute(s);
                }
                std::memcpy(digests + (r0 + l) * 8, s, 32);
            }
        }
    }
}
// out[i] = compress(a[i], b[i]) with a[i] = left + 8 i stride_a words ..., i in [lo, hi)
void compress_pairs(const uint32_t* left, size_t lstride, const uint32_t* right, size_t rstride, size_t lo, size_t hi, uint32_t* out) {
    const bool wide = p2x16_available();
    for (size_t i0 = lo; i0 < hi; i0 += 16) {
        const size_t cnt = std::min<size_t>(16, hi - i0);
        if (wide) {
            alignas(64) uint32_t st[16][16];
            std::memset(st, 0, sizeof st);
            for (size_t l = 0; l < cnt; l++)
                for (int k = 0; k < 8; k++) { st[k][l] = left[(i0 + l) * lstride + k]; st[8 + k][l] = right[(i0 + l) * rstride + k]; }
            p2x16_permute(st);
            for (size_t l = 0; l < cnt; l++) for (int k = 0; k < 8; k++) out[(i0 + l) * 8 + k] = st[k][l];
        } else {
            for (size_t l = 0; l < cnt; l++) p2_compress(left + (i0 + l) * lstride, right + (i0 + l) * rstride, out + (i0 + l) * 8);
        }
    }
}
}  // namespace

// root of the mixed-height commitment (op_merkle_commit_mixed, context.cpp): the tallest matrices form the leaves, a shorter one is
// injected at the level with as many nodes as it has rows -- node = compress(node, sponge(row)); matrices of one height in the given order
void host_merkle_root_mixed(const HostMat* mats, int nmats, uint32_t root[8]) {
    int log_h = 0;
    for (int k = 0; k < nmats; k++) log_h = std::max(log_h, mats[k].log_h);
    auto of_height = [&](int lh) { std::vector<HostMat> v; for (int k = 0; k < nmats; k++) if (mats[k].log_h == lh) v.push_back(mats[k]); return v; };
    size_t cnt = (size_t)1 << log_h;
    std::vector<uint32_t> level(cnt * 8), next, extra;
    {
        const std::vector<HostMat> top = of_height(log_h);
        parallel_for(cnt, 1024, [&](size_t lo, size_t hi) { sponge_rows(top.data(), (int)top.size(), lo, hi, level.data()); });
    }
    for (int lvl = log_h - 1; lvl >= 0; lvl--) {
        cnt >>= 1;
        next.assign(cnt * 8, 0u);
        parallel_for(cnt, 1024, [&](size_t lo, size_t hi) { compress_pairs(level.data(), 16, level.data() + 8, 16, lo, hi, next.data()); });
        const std::vector<HostMat> inj = of_height(lvl);
        if (!inj.empty()) {
            extra.assign(cnt * 8, 0u);
            parallel_for(cnt, 1024, [&](size_t lo, size_t hi) { sponge_rows(inj.data(), (int)inj.size(), lo, hi, extra.data()); });
            level.assign(cnt * 8, 0u);
            parallel_for(cnt, 1024, [&](size_t lo, size_t hi) { compress_pairs(next.data(), 8, extra.data(), 8, lo, hi, level.data()); });
        } else level.swap(next);
    }
    std::memcpy(root, level.data(), 32);
}

// The key of a machine from its preprocessed traces on the host: traces[c] = [2^log_ns[c]][pre_widths[c]] Montgomery words (NULL / width 0:
// the chip has none), chips tallest first as zkhip_machine_setup takes them.  root_out: 8 CANONICAL words, what zkhip_machine_setup returns.
int host_machine_key_root(const uint32_t* const* traces, const int32_t* log_ns, const uint32_t* pre_widths, int n_chips, int log_blowup, uint32_t root_out[8]) {
    std::vector<std::vector<uint32_t>> ldes;
    std::vector<HostMat> mats;
    ldes.reserve((size_t)n_chips);
    for (int c = 0; c < n_chips; c++) {
        const uint32_t pw = pre_widths[c];
        if (!pw) continue;
        if (!traces[c]) return -1;
        ldes.emplace_back(((size_t)1 << (log_ns[c] + log_blowup)) * pw);
        host_coset_lde(traces[c], pw, log_ns[c], pw, log_blowup, MONTY_GEN, ldes.back().data());
        mats.push_back(HostMat{ldes.back().data(), pw, pw, log_ns[c] + log_blowup});
    }
    if (mats.empty()) return -1;
    uint32_t root_m[8];
    host_merkle_root_mixed(mats.data(), (int)mats.size(), root_m);
    for (int i = 0; i < 8; i++) root_out[i] = from_monty(root_m[i]);
    return 0;
}

}  // namespace zk

using namespace zk;

extern "C" {

// zkhip_machine_setup's root without a device: HOST traces in, the 8 canonical key words out (include/zkhip.h)
int zkhip_machine_key_host(const uint32_t* const* h_traces, const int32_t* log_ns, const uint32_t* pre_widths, int n_chips, const zkhip_params* prm, uint32_t root[8]) {
    try {
        if (!h_traces || !log_ns || !pre_widths || !prm || !root || n_chips < 1 || n_chips > 32) return fail(ZKHIP_ERR_INVALID, "machine_key_host: bad arguments");
        if (prm->log_blowup < 1 || prm->log_blowup > 3) return fail(ZKHIP_ERR_INVALID, "machine_key_host: log_blowup in [1,3]");
        bool any = false;
        for (int c = 0; c < n_chips; c++) {
            if (log_ns[c] < 5 || log_ns[c] > 22 || (c && log_ns[c] > log_ns[c - 1])) return fail(ZKHIP_ERR_INVALID, "machine_key_host: log_n in [5,22], tallest first");
            if (pre_widths[c] % 4 != 0 || pre_widths[c] > 1024) return fail(ZKHIP_ERR_INVALID, "machine_key_host: preprocessed width a multiple of 4 up to 1024 (0: none)");
            if (pre_widths[c] && !h_traces[c]) return fail(ZKHIP_ERR_INVALID, "machine_key_host: a chip with preprocessed columns needs its trace");
            any = any || pre_widths[c] != 0;
        }
        if (!any) return fail(ZKHIP_ERR_INVALID, "machine_key_host: no chip has preprocessed columns");
        if (host_machine_key_root(h_traces, log_ns, pre_widths, n_chips, prm->log_blowup, root) != 0) return fail(ZKHIP_ERR_INTERNAL, "machine_key_host: failed");
        return ZKHIP_OK;
    } catch (const std::bad_alloc&) {
        return fail(ZKHIP_ERR_NOMEM, "machine_key_host: out of host memory");
    } catch (...) {
        return fail(ZKHIP_ERR_INTERNAL, "machine_key_host: exception");
    }
}

}  // extern "C"
