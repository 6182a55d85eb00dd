// guest_prover_hip.cpp -- see guest_prover_hip.hpp.  Links against libzkhip.so only
// through its C ABI (include/zkhip.h), exactly as the Rust crate would.
#include "guest_prover_hip.hpp"

#include <cstdlib>
#include <array>
#include <atomic>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <stdexcept>

#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"
#include "../../include/zkhip_hal.h"

namespace zktls {

const char* prover_type_name(ProverType mode) {
    switch (mode) {
        case ProverType::Mock: return "mock";
        case ProverType::Local: return "local";
        case ProverType::Hip: return "hip";
        case ProverType::Network: return "network";
    }
    return "mock";
}

void set_env(ProverType mode) { setenv("SP1_PROVER", prover_type_name(mode), 1); }   // sp1.rs:23-27
void set_env_r0(ProverType mode) {                                                    // prover.rs:19-28
    switch (mode) {
        case ProverType::Mock: setenv("RISC0_DEV_MODE", "true", 1); break;
        case ProverType::Local:
        case ProverType::Hip: setenv("RISC0_PROVER", "local", 1); break;          // like the reference's Cuda arm
        case ProverType::Network: setenv("RISC0_PROVER", "bonsai", 1); break;
    }
}

// the digest lives in libzkhip (zkhip_request_digest) so that the Rust glue derives the very same words through its FFI
std::vector<uint32_t> request_digest(const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf) {
    std::vector<uint32_t> out(8);
    if (zkhip_request_digest(cbor.data(), cbor.size(), elf.data(), elf.size(), out.data()) != ZKHIP_OK)
        throw std::runtime_error(std::string("zkhip_request_digest: ") + zkhip_last_error());
    return out;
}

std::vector<uint8_t> pack_shard_proofs(const std::vector<std::vector<uint8_t>>& proofs, uint32_t flags) {
    std::vector<uint8_t> out;
    auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) out.push_back((uint8_t)(v >> (8 * i))); };
    put32(0x42544B5Au);   // "ZKTB"
    put32(2);
    put32(flags);
    put32((uint32_t)proofs.size());
    for (const auto& p : proofs) {
        put32((uint32_t)p.size());
        out.insert(out.end(), p.begin(), p.end());
    }
    return out;
}

bool unpack_shard_proofs(const std::vector<uint8_t>& blob, std::vector<std::vector<uint8_t>>* proofs, uint32_t* flags) {
    auto get32 = [&](size_t off) { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)blob[off + i] << (8 * i); return v; };
    if (blob.size() < 16 || get32(0) != 0x42544B5Au || get32(4) != 2) return false;
    if (flags) *flags = get32(8);
    const uint32_t n = get32(12);
    size_t off = 16;
    proofs->clear();
    for (uint32_t i = 0; i < n; i++) {
        if (off + 4 > blob.size()) return false;
        const uint32_t len = get32(off);
        off += 4;
        if (off + len > blob.size()) return false;
        proofs->emplace_back(blob.begin() + off, blob.begin() + off + len);
        off += len;
    }
    return off == blob.size();
}

namespace {
// A worker's context and trace buffer.  Creating and freeing a context (stream + multi-GiB workspaces) costs more than a proof,
// and a prover serves many requests, so finished workers park them in a process-wide pool (zktls_release_cached frees it).
struct Slot { int device; size_t trace_bytes; zkhip_ctx* ctx; void* d_trace; zkhip_machine_key* key; uint32_t key_root[8]; };
std::mutex g_slots_mu;
std::vector<Slot> g_slots;
// the compress stage's key (the shard verifier machine's preprocessed traces on the device, zkhip_shard_verifier_setup: 30 - 110 ms at the
// headline shape): a function of the shape, so a prover that serves many requests of one plan keeps it -- parked like the contexts
// (a machine key belongs to the context that made it: the pair is parked together)
struct JoinKey { int device; int32_t log_n; uint32_t width; int32_t queries, pow_bits; uint32_t join; zkhip_ctx* ctx; zkhip_machine_key* key; uint32_t vk[8];
                 uint64_t p2_generation; };          // the key commits with the Poseidon2 tables in effect when it was made (zkhip_load_poseidon2_params moves the counter)
std::vector<JoinKey> g_join_keys;           // (under g_slots_mu)
bool take_join_key(int device, const zktls::ShardPlan& plan, uint32_t join, JoinKey* out) {
    const uint64_t gen = zkhip_poseidon2_params_generation();
    std::vector<JoinKey> stale;
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        for (size_t i = 0; i < g_join_keys.size();) {
            const JoinKey& k = g_join_keys[i];
            if (k.p2_generation != gen) { stale.push_back(k); g_join_keys.erase(g_join_keys.begin() + (long)i); continue; }     // made under other tables: a verifier would derive another key
            if (!found && k.device == device && k.log_n == plan.log_n && k.width == plan.width && k.queries == plan.num_queries && k.pow_bits == plan.pow_bits && k.join == join) {
                *out = k;
                g_join_keys.erase(g_join_keys.begin() + (long)i);
                found = true;
                continue;
            }
            i++;
        }
    }
    for (auto& k : stale) { zkhip_machine_key_destroy(k.key); zkhip_ctx_destroy(k.ctx); }
    return found;
}
void park_join_key(const JoinKey& k) {
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        if (g_join_keys.size() < 4 && k.p2_generation == zkhip_poseidon2_params_generation()) { g_join_keys.push_back(k); return; }
    }
    zkhip_machine_key_destroy(k.key);
    zkhip_ctx_destroy(k.ctx);
}
// the pair taken for one compress stage: parked again when the stage went through, destroyed on every other way out (an exception included)
struct JoinGuard {
    JoinKey jk;
    bool done = false;
    ~JoinGuard() {
        if (!jk.ctx) return;
        (void)zkhip_ctx_sync(jk.ctx);
        if (done) { park_join_key(jk); return; }
        if (jk.key) zkhip_machine_key_destroy(jk.key);
        zkhip_ctx_destroy(jk.ctx);
    }
};
struct CtxGuard {
    int device = 0;
    size_t trace_bytes = 0;
    zkhip_ctx* ctx = nullptr;
    void* d_trace = nullptr;
    zkhip_machine_key* key = nullptr;   // the SHA-256 machine's proving key made on this context (setup), parked with it
    uint32_t key_root[8] = {0};
    bool healthy = false;               // set once the worker finished without an error: only then is the slot reused
    bool take(int dev, size_t bytes) {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        for (size_t i = 0; i < g_slots.size(); i++)
            if (g_slots[i].device == dev && g_slots[i].trace_bytes == bytes) {
                ctx = g_slots[i].ctx; d_trace = g_slots[i].d_trace; device = dev; trace_bytes = bytes;
                key = g_slots[i].key; std::memcpy(key_root, g_slots[i].key_root, sizeof key_root);
                g_slots.erase(g_slots.begin() + (long)i);
                return true;
            }
        return false;
    }
    ~CtxGuard() {
        if (healthy && ctx && d_trace) {
            std::lock_guard<std::mutex> lk(g_slots_mu);
            Slot s{device, trace_bytes, ctx, d_trace, key, {0}};
            std::memcpy(s.key_root, key_root, sizeof key_root);
            g_slots.push_back(s);
            return;
        }
        if (key) zkhip_machine_key_destroy(key);
        if (ctx && d_trace) zkhip_free(ctx, d_trace);
        if (ctx) zkhip_ctx_destroy(ctx);
    }
};
[[noreturn]] void fail_zkhip(const char* what) { throw std::runtime_error(std::string(what) + ": " + zkhip_last_error()); }
}  // namespace

// the join machine (zkhip_prove_shard_verifier over J shard proofs of the plan's shape) as the INNER machine of machine mode: its eight chips as the
// library describes them, the join key's root, how join proofs are made
namespace {
struct JoinMachine {
    std::vector<std::vector<uint32_t>> progs, tabs;
    std::vector<const uint32_t*> pp, tp;
    std::vector<size_t> pw_, tw_;
    int32_t lns[8]; uint32_t widths[8], pres[8];
    zkhip_machine_desc desc{};
    bool build(const ShardPlan& plan, uint32_t J, const uint32_t join_key[8]) {
        progs.resize(8); tabs.resize(8); pp.resize(8); tp.resize(8); pw_.resize(8); tw_.resize(8);
        for (int i = 0; i < 8; i++) {
            int ln = 0; uint32_t mw = 0, pw = 0;
            for (int kind = 0; kind < 2; kind++) {
                std::vector<uint32_t>& dst = kind ? tabs[(size_t)i] : progs[(size_t)i];
                const size_t n = zkhip_shard_verifier_describe(plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, J, i, kind, nullptr, 0, &ln, &mw, &pw);
                if (n == 0) return false;
                dst.resize(n);
                if (zkhip_shard_verifier_describe(plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, J, i, kind, dst.data(), n, &ln, &mw, &pw) != n) return false;
            }
            lns[i] = ln; widths[i] = mw; pres[i] = pw;
            pp[(size_t)i] = progs[(size_t)i].data(); tp[(size_t)i] = tabs[(size_t)i].data(); pw_[(size_t)i] = progs[(size_t)i].size(); tw_[(size_t)i] = tabs[(size_t)i].size();
        }
        desc.n_chips = 8; desc.log_ns = lns; desc.widths = widths; desc.pre_widths = pres; desc.programs = pp.data(); desc.program_words = pw_.data();
        desc.tables = tp.data(); desc.table_words = tw_.data();
        std::memcpy(desc.key_root, join_key, 32);
        desc.num_queries = plan.num_queries; desc.pow_bits = plan.pow_bits; desc.n_public = 9u * J;
        return true;
    }
};
std::atomic<uint32_t> g_join_size{0};
}  // namespace

// core -> COMPRESS (sp1.rs:116; sp1-cuda's prove_core -> compress): `proofs` = the plan's shard proofs, public values request digest | shard index
static std::vector<uint8_t> compress_shard_proofs(const std::vector<int>& devices, const ShardPlan& plan, const zkhip_params& prm, const std::vector<uint32_t>& digest,
                                                  const std::vector<std::vector<uint8_t>>& proofs) {
    if (devices.empty()) throw std::runtime_error("device list is empty");
    const int device = devices[0];
    // core -> COMPRESS: the shard proofs verified in-circuit by ONE proof (the shard verifier machine, csrc/shard_verifier.inl); the blob then
    // carries that proof and the key of the shape instead of the shard proofs
    // more shards than one join holds: several joins of ONE shape (compress_join_size: the last one repeats the execution's last shard proof, so that
    // every join has the same key); the blob carries them in shard order
    const uint32_t J = compress_join_size(plan), n_joins = (plan.shards + J - 1) / J;
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    JoinGuard jg{JoinKey{device, plan.log_n, plan.width, plan.num_queries, plan.pow_bits, J, nullptr, nullptr, {0}, zkhip_poseidon2_params_generation()}};
    JoinKey& jk = jg.jk;
    if (!take_join_key(device, plan, J, &jk)) {
        if (zkhip_ctx_create(device, nullptr, &jk.ctx) != ZKHIP_OK) { jk.ctx = nullptr; fail_zkhip("zkhip_ctx_create"); }
    }
    // (one join: this context proves it and keeps the shape's key, parked with it; several: the batch entry's pooled contexts make the joins and keep
    // their own keys -- this context then proves the top only, and its key slot stays empty until a one-join request of the same shape comes by)
    if (n_joins == 1 && !jk.key &&
        zkhip_shard_verifier_setup(jk.ctx, plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, J, &outer, &jk.key, jk.vk) != ZKHIP_OK) {
        jk.key = nullptr;
        throw std::runtime_error(std::string("zkhip_shard_verifier_setup: ") + zkhip_last_error());
    }
    zkhip_ctx* const jctx = jk.ctx;
    zkhip_machine_key* key = jk.key;
    uint32_t vk[8];
    std::memcpy(vk, jk.vk, sizeof vk);
    const size_t jcap = zkhip_shard_verifier_proof_size(plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, J, &outer);
    std::vector<std::vector<uint8_t>> entries;
    if (n_joins > 1) {
        // several joins of one shape: ONE call deals them over the prover's devices, four in flight on each (zkhip_prove_shard_verifier_batch: pooled contexts
        // that keep the shape's key); every join is checked on its worker's thread (sp1.rs:120: the prover checks its own proof)
        std::vector<const uint8_t*> ptrs((size_t)n_joins * J);
        std::vector<size_t> lens((size_t)n_joins * J), jlens(n_joins);
        std::vector<uint32_t> pvs;
        for (uint32_t c = 0; c < n_joins; c++)
            for (uint32_t k = 0; k < J; k++) {
                const uint32_t sidx = c * J + k < plan.shards ? c * J + k : plan.shards - 1;
                ptrs[(size_t)c * J + k] = proofs[sidx].data(); lens[(size_t)c * J + k] = proofs[sidx].size();
                pvs.insert(pvs.end(), digest.begin(), digest.end());
                pvs.push_back(sidx);
            }
        std::vector<uint8_t> joined((size_t)n_joins * jcap);
        uint32_t bvk[8];
        if (zkhip_prove_shard_verifier_batch(devices.data(), (int)devices.size(), ptrs.data(), lens.data(), (size_t)n_joins * J, J, plan.log_n, plan.width, pvs.data(), 9, &prm, &outer,
                                             0, 1, joined.data(), jcap, jlens.data(), bvk) != ZKHIP_OK)
            throw std::runtime_error(std::string("zkhip_prove_shard_verifier_batch: ") + zkhip_last_error());
        if (jk.key && std::memcmp(bvk, vk, 32) != 0) throw std::runtime_error("compress: the batch's key differs from the shape's");
        std::memcpy(vk, bvk, 32);
        std::memcpy(jk.vk, bvk, 32);
        for (uint32_t c = 0; c < n_joins; c++) entries.emplace_back(joined.begin() + (long)((size_t)c * jcap), joined.begin() + (long)((size_t)c * jcap + jlens[c]));
    }
    if (n_joins == 1) {
        const uint32_t c = 0;
        std::vector<const uint8_t*> ptrs(J);
        std::vector<size_t> lens(J);
        std::vector<uint32_t> pvs;
        for (uint32_t k = 0; k < J; k++) {
            const uint32_t sidx = c * J + k < plan.shards ? c * J + k : plan.shards - 1;
            ptrs[k] = proofs[sidx].data(); lens[k] = proofs[sidx].size();
            pvs.insert(pvs.end(), digest.begin(), digest.end());
            pvs.push_back(sidx);
        }
        std::vector<uint8_t> joined(jcap);
        size_t jlen = 0;
        const int rc = zkhip_prove_shard_verifier(jctx, key, ptrs.data(), lens.data(), J, plan.log_n, plan.width, pvs.data(), 9, &prm, &outer, joined.data(), jcap, &jlen);
        if (rc != ZKHIP_OK) throw std::runtime_error(std::string("zkhip_prove_shard_verifier: ") + zkhip_last_error());
        joined.resize(jlen);
        int reason = 0;
        if (zkhip_verify_shard_recursive(joined.data(), jlen, plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, pvs.data(), 9, J, vk, &outer, &reason) != ZKHIP_OK) {
            throw std::runtime_error(std::string("zkhip_verify_shard_recursive: ") + zkhip_last_error());     // sp1.rs:120: the prover checks its own proof
        }
        entries.push_back(std::move(joined));
    }
    uint32_t flags_extra = 0;
    JoinMachine jm;
    // (a top the machine refuses -- thousands of public values per join: its transcript table runs out of columns -- leaves the joins side by side)
    if (n_joins > 1 && jm.build(plan, J, vk) && zkhip_machine_verifier_proof_size(&jm.desc, n_joins, &outer) != 0) {
        // THE TREE: the joins themselves are verified in-circuit by ONE proof (machine mode: csrc/machine_verifier.inl) -- the blob carries that
        // proof alone; its key is a function of (the join machine, the number of joins) and is derived by whoever checks the blob
        struct KeyGuard { zkhip_machine_key* k = nullptr; zkhip_ctx* c = nullptr; ~KeyGuard() { if (k) { (void)zkhip_ctx_sync(c); zkhip_machine_key_destroy(k); } } } tk;
        tk.c = jctx;
        uint32_t tvk[8];
        if (zkhip_machine_verifier_setup(jctx, &jm.desc, n_joins, &outer, &tk.k, tvk) != ZKHIP_OK) throw std::runtime_error(std::string("zkhip_machine_verifier_setup: ") + zkhip_last_error());
        const size_t tcap = zkhip_machine_verifier_proof_size(&jm.desc, n_joins, &outer);
        std::vector<uint8_t> top(tcap);
        std::vector<const uint8_t*> ptrs(n_joins);
        std::vector<size_t> lens(n_joins);
        std::vector<uint32_t> pvs;
        for (uint32_t c = 0; c < n_joins; c++) {
            ptrs[c] = entries[c].data(); lens[c] = entries[c].size();
            for (uint32_t k = 0; k < J; k++) {
                const uint32_t sidx = c * J + k < plan.shards ? c * J + k : plan.shards - 1;
                pvs.insert(pvs.end(), digest.begin(), digest.end());
                pvs.push_back(sidx);
            }
        }
        size_t tlen = 0;
        if (zkhip_prove_machine_verifier(jctx, tk.k, &jm.desc, ptrs.data(), lens.data(), n_joins, pvs.data(), 9u * J, &outer, top.data(), tcap, &tlen) != ZKHIP_OK)
            throw std::runtime_error(std::string("zkhip_prove_machine_verifier: ") + zkhip_last_error());
        top.resize(tlen);
        int reason = 0;
        if (zkhip_verify_machine_recursive(&jm.desc, top.data(), tlen, pvs.data(), 9u * J, n_joins, tvk, &outer, &reason) != ZKHIP_OK)
            throw std::runtime_error(std::string("zkhip_verify_machine_recursive: ") + zkhip_last_error());
        entries.clear();
        entries.push_back(std::move(top));
        flags_extra = BATCH_FLAG_TREE;
    }
    jg.done = true;                                  // (only after every join went through: every other way out destroys the pair -- JoinGuard)
    std::vector<uint8_t> tail(36);
    std::memcpy(tail.data(), vk, 32);
    const uint32_t cnt = plan.shards;
    std::memcpy(tail.data() + 32, &cnt, 4);
    entries.push_back(tail);
    return pack_shard_proofs(entries, BATCH_FLAG_SYNTHETIC | BATCH_FLAG_COMPRESSED | flags_extra);
}

// ---- synthetic shards in SP1's shard STRUCTURE (MachinePlan): the keyed machine, its shard proofs, the machine-mode compress stage ----
namespace {
constexpr uint32_t LKUP_MAGIC = 0x50554B4Cu, LKUP_SEND = 0u, LKUP_RECEIVE = 1u, LKUP_ONE = 0xFFFFFFFFu;
constexpr uint32_t BUS_SP1 = 300u;             // pair q of chip c travels on bus BUS_SP1 + 16 c + q (zktls_amd.device.Sp1ShapedShard: the same machine from Python)
constexpr uint64_t KEY_SHARD = 9999u;          // the preprocessed columns come from a stream of their own
constexpr uint32_t MACHINE_PUBLICS = 9u;       // request digest | shard index

// a machine as zkhip_machine_desc wants it, with the storage behind the pointers
struct MachineHolder {
    std::vector<std::vector<uint32_t>> progs, tabs;
    std::vector<const uint32_t*> pp, tp;
    std::vector<size_t> pw_, tw_;
    std::vector<int32_t> lns;
    std::vector<uint32_t> widths, pres;          // main widths, preprocessed widths
    zkhip_machine_desc desc{};
    void finish(const uint32_t key_root[8], int q, int pb, uint32_t n_public) {
        const size_t n = progs.size();
        pp.resize(n); tp.resize(n); pw_.resize(n); tw_.resize(n);
        for (size_t i = 0; i < n; i++) {
            pp[i] = progs[i].data(); pw_[i] = progs[i].size();
            tp[i] = tabs[i].empty() ? nullptr : tabs[i].data(); tw_[i] = tabs[i].size();
        }
        desc.n_chips = (int32_t)n; desc.log_ns = lns.data(); desc.widths = widths.data(); desc.pre_widths = pres.data();
        desc.programs = pp.data(); desc.program_words = pw_.data(); desc.tables = tp.data(); desc.table_words = tw_.data();
        if (key_root) std::memcpy(desc.key_root, key_root, 32);
        desc.num_queries = q; desc.pow_bits = pb; desc.n_public = n_public;
    }
    // the shard machine of a plan: every chip under the synthetic AIR as a constraint program over [preprocessed | main], its pairs as an interaction table
    void build_plan(const MachinePlan& plan, const uint32_t key_root[8]) {
        const size_t n = plan.chips.size();
        if (n == 0 || n > 16) throw std::runtime_error("machine plan: 1 to 16 chips");
        progs.assign(n, {}); tabs.assign(n, {}); lns.resize(n); widths.resize(n); pres.resize(n);
        for (size_t c = 0; c < n; c++) {
            const ChipPlan& ch = plan.chips[c];
            if (c && ch.log_n > plan.chips[c - 1].log_n) throw std::runtime_error("machine plan: chips tallest first");
            if (ch.width == 0 || ch.width % 4 || 8u * ch.pairs > ch.width) throw std::runtime_error("machine plan: a chip's width is a multiple of 4 and holds 8 columns per LogUp pair");
            if (ch.pre_width && (ch.pre_width % 8 || ch.pre_width >= ch.width || ch.partner >= 0)) throw std::runtime_error("machine plan: preprocessed columns are whole in-table pairs (a multiple of 8, fewer than the width)");
            if (ch.partner >= 0) {
                if ((size_t)ch.partner >= n || (size_t)ch.partner == c) throw std::runtime_error("machine plan: partner out of range");
                const ChipPlan& o = plan.chips[(size_t)ch.partner];
                if (o.partner != (int)c || o.log_n != ch.log_n || o.pairs != ch.pairs) throw std::runtime_error("machine plan: partners are mutual, of one height and one pair count");
            }
            size_t words = 0;
            if (zkhip_air_synthetic(ch.width, MACHINE_PUBLICS, nullptr, 0, &words) != ZKHIP_OK || words == 0) fail_zkhip("zkhip_air_synthetic");
            progs[c].resize(words);
            if (zkhip_air_synthetic(ch.width, MACHINE_PUBLICS, progs[c].data(), words, &words) != ZKHIP_OK) fail_zkhip("zkhip_air_synthetic");
            if (ch.pairs) {
                std::vector<uint32_t>& t = tabs[c];
                t = {LKUP_MAGIC, 2u * ch.pairs, 0u};
                for (uint32_t q = 0; q < ch.pairs; q++) {
                    const uint32_t to = ch.partner < 0 ? (uint32_t)c : (uint32_t)ch.partner;
                    const uint32_t send[] = {LKUP_SEND, LKUP_ONE, BUS_SP1 + 16u * (uint32_t)c + q, 2u, 8u * q, 8u * q + 1u};
                    const uint32_t recv[] = {LKUP_RECEIVE, LKUP_ONE, BUS_SP1 + 16u * to + q, 2u, 8u * q + 4u, 8u * q + 5u};
                    t.insert(t.end(), send, send + 6);
                    t.insert(t.end(), recv, recv + 6);
                }
                t[2] = (uint32_t)t.size();
            }
            lns[c] = ch.log_n; widths[c] = ch.width - ch.pre_width; pres[c] = ch.pre_width;
        }
        bool keyed = false;
        for (const ChipPlan& ch : plan.chips) keyed = keyed || ch.pre_width != 0;
        if (!keyed) throw std::runtime_error("machine plan: at least one chip carries preprocessed columns (the shards are proofs of a KEYED machine: what setup commits)");
        finish(key_root, plan.num_queries, plan.pow_bits, MACHINE_PUBLICS);
    }
    // the machine-mode machine over n_proofs proofs of `inner` as an inner machine itself (the library describes its chips): what a tree's top verifies
    void build_described(const zkhip_machine_desc& inner, size_t n_proofs, const uint32_t key_root[8], int q, int pb, uint32_t n_public) {
        progs.clear(); tabs.clear(); lns.clear(); widths.clear(); pres.clear();
        for (int i = 0; i < 16; i++) {
            int ln = 0; uint32_t mw = 0, pw = 0;
            const size_t np = zkhip_machine_verifier_describe(&inner, n_proofs, i, 0, nullptr, 0, &ln, &mw, &pw);
            if (np == 0) break;
            std::vector<uint32_t> prog(np);
            if (zkhip_machine_verifier_describe(&inner, n_proofs, i, 0, prog.data(), np, &ln, &mw, &pw) != np) fail_zkhip("zkhip_machine_verifier_describe");
            const size_t nt = zkhip_machine_verifier_describe(&inner, n_proofs, i, 1, nullptr, 0, &ln, &mw, &pw);
            std::vector<uint32_t> tab(nt);
            if (nt && zkhip_machine_verifier_describe(&inner, n_proofs, i, 1, tab.data(), nt, &ln, &mw, &pw) != nt) fail_zkhip("zkhip_machine_verifier_describe");
            progs.push_back(std::move(prog)); tabs.push_back(std::move(tab)); lns.push_back(ln); widths.push_back(mw); pres.push_back(pw);
        }
        if (progs.empty()) fail_zkhip("zkhip_machine_verifier_describe");
        finish(key_root, q, pb, n_public);
    }
};

uint64_t stream_seed(const std::vector<uint32_t>& digest) {
    uint64_t seed = 0;
    for (int i = 0; i < 4; i++) seed = (seed << 16) ^ digest[(size_t)i];
    return seed;
}
size_t plan_trace_bytes(const MachinePlan& plan, std::vector<size_t>* offsets) {
    size_t at = 0;
    for (const ChipPlan& ch : plan.chips) {
        if (offsets) offsets->push_back(at);
        at += ((((size_t)ch.width << ch.log_n) * 4u + 255u) / 256u) * 256u;
    }
    return at + 64u;                           // (+ 64: never the size of a single-matrix plan's buffer, whose parked slots these must not be taken for)
}
// setup (sp1.rs:113) on one context: the preprocessed columns of the plan's chips from the PROGRAM's stream, committed once
void machine_setup_on(zkhip_ctx* ctx, const MachinePlan& plan, const std::vector<uint8_t>& elf, const zkhip_params& prm, zkhip_machine_key** key, uint32_t root[8]) {
    const uint64_t key_seed = stream_seed(request_digest({}, elf));
    const size_t n = plan.chips.size();
    std::vector<zkhip_chip> pre(n);
    struct Bufs { zkhip_ctx* c; std::vector<void*> p; ~Bufs() { (void)zkhip_ctx_sync(c); for (void* q : p) zkhip_free(c, q); } } bufs{ctx, {}};
    for (size_t c = 0; c < n; c++) {
        const ChipPlan& ch = plan.chips[c];
        pre[c] = zkhip_chip{nullptr, 0, ch.log_n, 0, 0, -1};
        if (!ch.pre_width) continue;
        void* d = nullptr;
        if (zkhip_malloc(ctx, ((size_t)ch.width << ch.log_n) * 4u, &d) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
        bufs.p.push_back(d);
        if (zkhip_gen_trace_logup(ctx, key_seed, 100u * KEY_SHARD + c, ch.log_n, ch.width, (int)ch.pairs, (uint32_t*)d, ch.width) != ZKHIP_OK) fail_zkhip("zkhip_gen_trace_logup");
        pre[c] = zkhip_chip{(const uint32_t*)d, ch.width, ch.log_n, ch.pre_width, 0, -1};
    }
    if (zkhip_machine_setup(ctx, pre.data(), (int)n, &prm, key, root) != ZKHIP_OK) { *key = nullptr; fail_zkhip("zkhip_machine_setup"); }
}
std::vector<uint32_t> machine_publics(const std::vector<uint32_t>& digest, uint32_t J, uint32_t n_joins, uint32_t shards) {
    std::vector<uint32_t> pvs;
    for (uint32_t c = 0; c < n_joins; c++)
        for (uint32_t k = 0; k < J; k++) {
            const uint32_t sidx = c * J + k < shards ? c * J + k : shards - 1;       // the last join repeats the last shard (one shape, one key)
            pvs.insert(pvs.end(), digest.begin(), digest.end());
            pvs.push_back(sidx);
        }
    return pvs;
}
}  // namespace

namespace {
// A recursion machine's key is a function of (the inner machine's description, the number of proofs, the outer shape, the Poseidon2 tables): deriving it on the host
// commits the machine's preprocessed traces (hundreds of milliseconds at the bench's shapes), so a verifier that checks many blobs of one plan derives it ONCE --
// the cache is keyed by the description's exact bytes, nothing of a blob enters it
std::mutex g_derived_mu;
std::vector<std::pair<std::string, std::array<uint32_t, 8>>> g_derived;
std::string desc_bytes(const zkhip_machine_desc& d, size_t n_proofs, const zkhip_params& outer) {
    std::string b;
    auto put = [&](const void* p, size_t n) { b.append((const char*)p, n); };
    const uint64_t gen = zkhip_poseidon2_params_generation(), np = n_proofs;
    put(&gen, 8); put(&np, 8); put(&outer, sizeof outer);
    put(&d.n_chips, 4); put(d.key_root, 32); put(&d.num_queries, 4); put(&d.pow_bits, 4); put(&d.n_public, 4);
    for (int c = 0; c < d.n_chips; c++) {
        put(&d.log_ns[c], 4); put(&d.widths[c], 4); put(&d.pre_widths[c], 4);
        const uint64_t pw = d.program_words[c], tw = d.tables[c] ? d.table_words[c] : 0;
        put(&pw, 8); put(d.programs[c], pw * 4);
        put(&tw, 8); if (tw) put(d.tables[c], tw * 4);
    }
    return b;
}
bool machine_verifier_key_host_cached(const zkhip_machine_desc& d, size_t n_proofs, const zkhip_params& outer, uint32_t key[8]) {
    const std::string id = desc_bytes(d, n_proofs, outer);
    {
        std::lock_guard<std::mutex> lk(g_derived_mu);
        for (auto& e : g_derived)
            if (e.first == id) { std::memcpy(key, e.second.data(), 32); return true; }
    }
    if (zkhip_machine_verifier_key_host(&d, n_proofs, &outer, key) != ZKHIP_OK) return false;
    std::array<uint32_t, 8> k;
    std::memcpy(k.data(), key, 32);
    std::lock_guard<std::mutex> lk(g_derived_mu);
    if (g_derived.size() >= 8) g_derived.erase(g_derived.begin());
    g_derived.emplace_back(id, k);
    return true;
}

// the machine-mode join's proving key (the join machine's preprocessed traces on the device) with the context that made it: a function of (plan, the shard machine's
// key, J), parked between requests like the single-matrix path's JoinKey
struct MachineJoinKey { int device; std::string id; zkhip_ctx* ctx; zkhip_machine_key* key; uint32_t vk[8]; };
std::vector<MachineJoinKey> g_machine_join_keys;       // (under g_slots_mu)
bool take_machine_join_key(int device, const std::string& id, MachineJoinKey* out) {
    std::lock_guard<std::mutex> lk(g_slots_mu);
    for (size_t i = 0; i < g_machine_join_keys.size(); i++)
        if (g_machine_join_keys[i].device == device && g_machine_join_keys[i].id == id) {
            *out = g_machine_join_keys[i];
            g_machine_join_keys.erase(g_machine_join_keys.begin() + (long)i);
            return true;
        }
    return false;
}
void park_machine_join_key(const MachineJoinKey& k) {
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        if (g_machine_join_keys.size() < 2) { g_machine_join_keys.push_back(k); return; }
    }
    zkhip_machine_key_destroy(k.key);
    zkhip_ctx_destroy(k.ctx);
}
}  // namespace

MachinePlan MachinePlan::sp1_shaped(uint32_t shards) {
    MachinePlan p;
    p.chips = {ChipPlan{20, 96, 3, 1, 0}, ChipPlan{20, 32, 3, 0, 0}, ChipPlan{19, 64, 2, -1, 0}, ChipPlan{18, 128, 4, -1, 0}, ChipPlan{16, 256, 8, -1, 32}, ChipPlan{14, 40, 1, -1, 0}};
    p.shards = shards;
    return p;
}

uint32_t machine_join_size(const MachinePlan& plan) {
    const uint32_t shards = plan.shards ? plan.shards : 1u;
    MachineHolder m;
    const uint32_t zero[8] = {0};
    m.build_plan(plan, zero);
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    uint32_t most = 1;                                  // (a machine the join refuses at one proof: the join itself will say so)
    for (uint32_t n = 64; n >= 1; n--)                  // machine mode: at most 64 proofs, and a Poseidon2 chip of at most 2^22 rows
        if (zkhip_machine_verifier_proof_size(&m.desc, n, &outer) != 0) { most = n; break; }
    const uint32_t cap = g_join_size.load();
    if (cap && cap < most) most = cap;
    if (shards <= most) return shards;
    const uint32_t joins = (shards + most - 1u) / most;
    return (shards + joins - 1u) / joins;
}

// core -> COMPRESS for machine shards: the shard proofs verified in-circuit in machine mode; several joins are joined again
static std::vector<uint8_t> compress_machine_proofs(int device, const MachinePlan& plan, const uint32_t key_root[8], const std::vector<uint32_t>& digest,
                                                    const std::vector<std::vector<uint8_t>>& proofs) {
    const uint32_t J = machine_join_size(plan), n_joins = (plan.shards + J - 1) / J;
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    MachineHolder m;
    m.build_plan(plan, key_root);
    const size_t jcap = zkhip_machine_verifier_proof_size(&m.desc, J, &outer);
    if (jcap == 0) throw std::runtime_error(std::string("zkhip_machine_verifier_proof_size: ") + zkhip_last_error());
    struct Guard {
        zkhip_ctx* ctx = nullptr; zkhip_machine_key* jkey = nullptr; zkhip_machine_key* tkey = nullptr;
        int device = 0; std::string id; uint32_t vk[8] = {0}; bool done = false;
        ~Guard() {
            if (!ctx) return;
            (void)zkhip_ctx_sync(ctx);
            if (tkey) zkhip_machine_key_destroy(tkey);
            if (done && jkey) {                           // the stage went through: the pair is parked for the next request of this plan
                MachineJoinKey k{device, id, ctx, jkey, {0}};
                std::memcpy(k.vk, vk, 32);
                park_machine_join_key(k);
                return;
            }
            if (jkey) zkhip_machine_key_destroy(jkey);
            zkhip_ctx_destroy(ctx);
        }
    } g;
    g.device = device;
    g.id = desc_bytes(m.desc, J, outer);
    uint32_t jvk[8];
    MachineJoinKey parked{};
    if (take_machine_join_key(device, g.id, &parked)) {
        g.ctx = parked.ctx; g.jkey = parked.key;
        std::memcpy(jvk, parked.vk, 32);
    } else {
        if (zkhip_ctx_create(device, nullptr, &g.ctx) != ZKHIP_OK) { g.ctx = nullptr; fail_zkhip("zkhip_ctx_create"); }
        if (zkhip_machine_verifier_setup(g.ctx, &m.desc, J, &outer, &g.jkey, jvk) != ZKHIP_OK) { g.jkey = nullptr; fail_zkhip("zkhip_machine_verifier_setup"); }
    }
    std::memcpy(g.vk, jvk, 32);
    const std::vector<uint32_t> pvs = machine_publics(digest, J, n_joins, plan.shards);
    std::vector<std::vector<uint8_t>> entries;
    for (uint32_t c = 0; c < n_joins; c++) {
        std::vector<const uint8_t*> ptrs(J);
        std::vector<size_t> lens(J);
        for (uint32_t k = 0; k < J; k++) {
            const uint32_t sidx = c * J + k < plan.shards ? c * J + k : plan.shards - 1;
            ptrs[k] = proofs[sidx].data(); lens[k] = proofs[sidx].size();
        }
        std::vector<uint8_t> joined(jcap);
        size_t jlen = 0;
        const uint32_t* pv = pvs.data() + (size_t)c * J * MACHINE_PUBLICS;
        if (zkhip_prove_machine_verifier(g.ctx, g.jkey, &m.desc, ptrs.data(), lens.data(), J, pv, MACHINE_PUBLICS, &outer, joined.data(), jcap, &jlen) != ZKHIP_OK)
            fail_zkhip("zkhip_prove_machine_verifier");
        joined.resize(jlen);
        int reason = 0;
        if (zkhip_verify_machine_recursive(&m.desc, joined.data(), jlen, pv, MACHINE_PUBLICS, J, jvk, &outer, &reason) != ZKHIP_OK)      // sp1.rs:120: the prover checks its own proof
            fail_zkhip("zkhip_verify_machine_recursive");
        entries.push_back(std::move(joined));
    }
    uint32_t flags_extra = 0;
    if (n_joins > 1) {
        MachineHolder jm;
        jm.build_described(m.desc, J, jvk, plan.num_queries, plan.pow_bits, MACHINE_PUBLICS * J);
        const size_t tcap = zkhip_machine_verifier_proof_size(&jm.desc, n_joins, &outer);
        if (tcap != 0) {                                  // (a top the machine refuses leaves the joins side by side)
            uint32_t tvk[8];
            if (zkhip_machine_verifier_setup(g.ctx, &jm.desc, n_joins, &outer, &g.tkey, tvk) != ZKHIP_OK) { g.tkey = nullptr; fail_zkhip("zkhip_machine_verifier_setup"); }
            std::vector<const uint8_t*> ptrs(n_joins);
            std::vector<size_t> lens(n_joins);
            for (uint32_t c = 0; c < n_joins; c++) { ptrs[c] = entries[c].data(); lens[c] = entries[c].size(); }
            std::vector<uint8_t> top(tcap);
            size_t tlen = 0;
            if (zkhip_prove_machine_verifier(g.ctx, g.tkey, &jm.desc, ptrs.data(), lens.data(), n_joins, pvs.data(), MACHINE_PUBLICS * J, &outer, top.data(), tcap, &tlen) != ZKHIP_OK)
                fail_zkhip("zkhip_prove_machine_verifier (top)");
            top.resize(tlen);
            int reason = 0;
            if (zkhip_verify_machine_recursive(&jm.desc, top.data(), tlen, pvs.data(), MACHINE_PUBLICS * J, n_joins, tvk, &outer, &reason) != ZKHIP_OK)
                fail_zkhip("zkhip_verify_machine_recursive (top)");
            entries.clear();
            entries.push_back(std::move(top));
            flags_extra = BATCH_FLAG_TREE;
        }
    }
    g.done = true;
    std::vector<uint8_t> tail(36);
    std::memcpy(tail.data(), jvk, 32);
    const uint32_t cnt = plan.shards;
    std::memcpy(tail.data() + 32, &cnt, 4);
    entries.push_back(tail);
    return pack_shard_proofs(entries, BATCH_FLAG_SYNTHETIC | BATCH_FLAG_MACHINE | BATCH_FLAG_COMPRESSED | flags_extra);
}

int verify_machine_blob(const std::vector<uint8_t>& blob, const MachinePlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf,
                        const std::vector<uint8_t>& vk, int* reason) {
    try {
        std::vector<std::vector<uint8_t>> entries;
        uint32_t flags = 0;
        if (vk.size() != 64 || plan.shards == 0 || !unpack_shard_proofs(blob, &entries, &flags) || !(flags & BATCH_FLAG_MACHINE) || !(flags & BATCH_FLAG_SYNTHETIC)) return -1;
        const std::vector<uint32_t> pd = request_digest({}, elf);
        if (std::memcmp(vk.data() + 32, pd.data(), 32) != 0) return -1;                      // a key made for another program
        uint32_t root[8];
        std::memcpy(root, vk.data(), 32);
        const std::vector<uint32_t> digest = request_digest(cbor, elf);
        MachineHolder m;
        m.build_plan(plan, root);
        const zkhip_params prm{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
        int why = 0;
        if (!(flags & BATCH_FLAG_COMPRESSED)) {
            if (entries.size() != plan.shards) return -1;
            for (uint32_t s = 0; s < plan.shards; s++) {
                std::vector<uint32_t> pv(digest);
                pv.push_back(s);
                const int rc = zkhip_verify_machine_keyed(entries[s].data(), entries[s].size(), m.lns.data(), m.widths.data(), m.pres.data(), root, m.pp.data(), m.pw_.data(), m.tp.data(),
                                                          m.tw_.data(), m.desc.n_chips, pv.data(), pv.size(), &prm, &why);
                if (reason) *reason = why;
                if (rc != ZKHIP_OK) return -2;
            }
            return 0;
        }
        if (entries.size() < 2 || entries.back().size() != 36) return -1;
        uint32_t cnt = 0;
        std::memcpy(&cnt, entries.back().data() + 32, 4);
        if (cnt != plan.shards) return -1;
        const uint32_t J = machine_join_size(plan), n_joins = (cnt + J - 1) / J;
        uint32_t jvk[8];
        if (!machine_verifier_key_host_cached(m.desc, J, prm, jvk)) return -1;               // the join's key is DERIVED (once per plan and key): the blob's copy is informative
        const std::vector<uint32_t> pvs = machine_publics(digest, J, n_joins, cnt);
        if (flags & BATCH_FLAG_TREE) {
            if (n_joins < 2 || entries.size() != 2) return -1;
            MachineHolder jm;
            jm.build_described(m.desc, J, jvk, plan.num_queries, plan.pow_bits, MACHINE_PUBLICS * J);
            uint32_t tvk[8];
            if (!machine_verifier_key_host_cached(jm.desc, n_joins, prm, tvk)) return -1;
            const int rc = zkhip_verify_machine_recursive(&jm.desc, entries[0].data(), entries[0].size(), pvs.data(), MACHINE_PUBLICS * J, n_joins, tvk, &prm, &why);
            if (reason) *reason = why;
            return rc == ZKHIP_OK ? 0 : -2;
        }
        if (entries.size() != (size_t)n_joins + 1) return -1;
        for (uint32_t c = 0; c < n_joins; c++) {
            const int rc = zkhip_verify_machine_recursive(&m.desc, entries[c].data(), entries[c].size(), pvs.data() + (size_t)c * J * MACHINE_PUBLICS, MACHINE_PUBLICS, J, jvk, &prm, &why);
            if (reason) *reason = why;
            if (rc != ZKHIP_OK) return -2;
        }
        return 0;
    } catch (...) {
        return -1;
    }
}

ProveResult HipGuestProver::prove(const GuestInput& input, const std::vector<uint8_t>& guest_program) {
    if (backend_ == Backend::Risc0) set_env_r0(mode_);   // prover.rs:65
    else set_env(mode_);                              // sp1.rs:72
    const std::vector<uint8_t> elf = guest_program;   // sp1.rs:74: the ELF is copied once
    ProveResult r;
    try {                                             // sp1.rs:85: nothing may unwind past here
        r = prove_inner(input, elf);
    } catch (const std::exception& e) {
        r = ProveResult{};
        r.error = e.what();
    } catch (...) {
        r = ProveResult{};
        r.error = "unknown failure in HipGuestProver::prove";
    }
    if (r.ok && r.proof.size() <= 4) r.proof.clear();   // sp1.rs:128-130
    return r;
}

SetupResult HipGuestProver::setup(const std::vector<uint8_t>& guest_program) {
    SetupResult s;
    try {
        if (guest_program.empty()) throw std::runtime_error("guest program is empty");
        if (!commitment_ && !machine_) throw std::runtime_error("setup: only the input-commitment guest and machine shards have preprocessed tables (with_input_commitment(), with_synthetic_machine())");
        if (backend_ != Backend::Sp1) throw std::runtime_error("setup: the keyed machine uses the SP1 proof shape");
        if (mode_ == ProverType::Network) throw std::runtime_error("network proving is not provided by the HIP backend");
        std::vector<uint8_t> vk(64, 0);
        const std::vector<uint32_t> pd = request_digest({}, guest_program);
        std::memcpy(vk.data() + 32, pd.data(), 32);
        if (machine_) {
            // the machine's preprocessed columns, a function of (plan, program), committed on the first device: the key every shard proof opens against
            if (mode_ != ProverType::Mock) {
                if (devices_.empty()) throw std::runtime_error("device list is empty");
                const zkhip_params prm{1, mplan_.num_queries, mplan_.pow_bits, 0, 0, 0, 0, 0};
                MachineHolder shape;
                const uint32_t zero[8] = {0};
                shape.build_plan(mplan_, zero);                            // (refuses a malformed plan before any device work)
                struct G { zkhip_ctx* c = nullptr; zkhip_machine_key* k = nullptr; ~G() { if (k) { (void)zkhip_ctx_sync(c); zkhip_machine_key_destroy(k); } if (c) zkhip_ctx_destroy(c); } } g;
                if (zkhip_ctx_create(devices_[0], nullptr, &g.c) != ZKHIP_OK) { g.c = nullptr; fail_zkhip("zkhip_ctx_create"); }
                uint32_t root[8];
                machine_setup_on(g.c, mplan_, guest_program, prm, &g.k, root);
                std::memcpy(vk.data(), root, 32);
            }
            vk_ = vk;
            s.vk = vk;
            s.ok = true;
            return s;
        }
        if (mode_ != ProverType::Mock) {
            if (devices_.empty()) throw std::runtime_error("device list is empty");
            const zkhip_params prm{1, plan_.num_queries, plan_.pow_bits, 0, 0, 0, 0, 0};
            CtxGuard g;
            constexpr size_t PLACEHOLDER = 256;
            if (!g.take(devices_[0], PLACEHOLDER)) {
                g.device = devices_[0]; g.trace_bytes = PLACEHOLDER;
                if (zkhip_ctx_create(devices_[0], nullptr, &g.ctx) != ZKHIP_OK) fail_zkhip("zkhip_ctx_create");
                if (zkhip_malloc(g.ctx, PLACEHOLDER, &g.d_trace) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
            }
            if (!g.key && zkhip_sha256_setup(g.ctx, &prm, &g.key, g.key_root) != ZKHIP_OK) fail_zkhip("zkhip_sha256_setup");
            g.healthy = true;
            std::memcpy(vk.data(), g.key_root, 32);
        }
        vk_ = vk;
        s.vk = vk;
        s.ok = true;
    } catch (const std::exception& e) {
        s = SetupResult{};
        s.error = e.what();
    }
    return s;
}

static std::vector<uint8_t> length_entry(uint64_t n) {
    std::vector<uint8_t> e(8);
    for (int i = 0; i < 8; i++) e[(size_t)i] = (uint8_t)(n >> (8 * i));
    return e;
}
bool commitment_blob_length(const std::vector<uint8_t>& blob, uint64_t* message_len) {
    std::vector<std::vector<uint8_t>> proofs;
    uint32_t flags = 0;
    if (!unpack_shard_proofs(blob, &proofs, &flags) || proofs.size() < 2 || !(flags & BATCH_FLAG_INPUT_SHA256) || proofs.back().size() != 8) return false;
    uint64_t n = 0;
    for (int i = 0; i < 8; i++) n |= (uint64_t)proofs.back()[(size_t)i] << (8 * i);
    if (message_len) *message_len = n;
    return true;
}

int verify_commitment_blob(const std::vector<uint8_t>& blob, const std::vector<uint8_t>& output, const std::vector<uint8_t>& vk,
                           int num_queries, int pow_bits, int* reason, Backend backend) {
    std::vector<std::vector<uint8_t>> proofs;
    uint32_t flags = 0;
    if (reason) *reason = 0;
    if (!unpack_shard_proofs(blob, &proofs, &flags) || proofs.size() < 2 || !(flags & BATCH_FLAG_INPUT_SHA256) || output.size() != 32 || proofs.back().size() != 8) return -1;
    // the LAST entry is the statement's other half: the input's length in bytes (8 LE bytes).  The proofs say "output = SHA-256 of a message
    // of exactly this length" (the padding is constrained in-circuit); a consumer that holds the input compares the length it knows
    // (commitment_blob_length)
    uint64_t message_len = 0;
    for (int i = 0; i < 8; i++) message_len |= (uint64_t)proofs.back()[(size_t)i] << (8 * i);
    proofs.pop_back();
    // what the caller expects decides the path; the blob's flags must agree
    const bool want_keyed = !vk.empty();
    if (want_keyed && (vk.size() != 64 || !(flags & BATCH_FLAG_KEYED) || (flags & BATCH_FLAG_CHAINED))) { if (reason) *reason = 2; return -1; }
    if (!want_keyed && (flags & BATCH_FLAG_KEYED)) { if (reason) *reason = 2; return -1; }
    zkhip_params prm{1, num_queries, pow_bits, 0, 0, 0, 0, 0};
    if (backend == Backend::Risc0) {                     // the shape prove_inner uses for this backend; log_n is the proof's (header word 2)
        if ((flags & (BATCH_FLAG_CHAINED | BATCH_FLAG_KEYED)) || proofs[0].size() < 12) { if (reason) *reason = 2; return -1; }
        uint32_t log_n = 0;
        std::memcpy(&log_n, proofs[0].data() + 8, 4);
        if (log_n < 6 || log_n > 22) return -1;
        int lf = 8;
        while (lf > (int)log_n || ((int)log_n - lf) % 4 != 0) lf--;
        const bool defaults = num_queries == 100 && pow_bits == 16;
        prm = zkhip_params{2, defaults ? 50 : num_queries, defaults ? 0 : pow_bits, 0, 4, lf, 24, 0};
    }
    if ((flags & BATCH_FLAG_CHAINED) && (flags & BATCH_FLAG_COMPRESSED)) {
        // entry 0: the chaining values, entry 1: the ONE proof (the shards verified in-circuit), entry 2: the key the prover used.  The key
        // is a function of (shard count, shapes): derived HERE, on the host's cores -- the blob's copy only has to agree
        if (proofs.size() != 3 || proofs[2].size() != 32 || proofs[0].size() < 64 || proofs[0].size() % 32) return -1;
        const size_t n = proofs[0].size() / 32 - 1;
        if (zkhip_sha256_sharded_count((size_t)message_len, 14) != n) { if (reason) *reason = 1; return -1; }
        uint32_t key[8];
        if (zkhip_sha256_compress_key_host((size_t)message_len, 14, &prm, &prm, key) != ZKHIP_OK) return -1;
        if (std::memcmp(key, proofs[2].data(), 32) != 0) { if (reason) *reason = 2; return -1; }
        std::vector<uint32_t> chain((n + 1) * 8);
        std::memcpy(chain.data(), proofs[0].data(), proofs[0].size());
        return zkhip_verify_sha256_compressed(proofs[1].data(), proofs[1].size(), output.data(), message_len, chain.data(), 14, key, &prm, &prm, reason);
    }
    if (flags & BATCH_FLAG_COMPRESSED) { if (reason) *reason = 2; return -1; }
    if (flags & BATCH_FLAG_CHAINED) {                    // entry 0: the chaining values; entries 1..n: the shard proofs
        const size_t n = proofs.size() - 1;
        if (n < 1 || proofs[0].size() != (n + 1) * 32) return -1;
        size_t stride = 0;
        for (size_t s = 1; s <= n; s++) stride = std::max(stride, proofs[s].size());
        std::vector<uint8_t> buf(n * stride);
        std::vector<size_t> lens(n);
        for (size_t s = 0; s < n; s++) { std::memcpy(buf.data() + s * stride, proofs[s + 1].data(), proofs[s + 1].size()); lens[s] = proofs[s + 1].size(); }
        std::vector<uint32_t> chain((n + 1) * 8);
        std::memcpy(chain.data(), proofs[0].data(), proofs[0].size());
        size_t bad = 0;
        return zkhip_verify_sha256_sharded(buf.data(), stride, lens.data(), n, chain.data(), 14, output.data(), message_len, &prm, &bad, reason);
    }
    if (proofs.size() != 1) return -1;
    if (want_keyed) return zkhip_verify_sha256_machine(proofs[0].data(), proofs[0].size(), output.data(), message_len, (const uint32_t*)vk.data(), &prm, reason);
    return zkhip_verify_sha256(proofs[0].data(), proofs[0].size(), output.data(), message_len, &prm, reason);
}

ProveResult HipGuestProver::prove_inner(const GuestInput& input, const std::vector<uint8_t>& elf) {
    ProveResult r;
    if (elf.empty()) throw std::runtime_error("guest program is empty");
    const std::vector<uint32_t> digest = request_digest(input.cbor, elf);
    r.output.resize(32);
    std::memcpy(r.output.data(), digest.data(), 32);
    if (mode_ == ProverType::Mock) {                  // executes nothing, returns a placeholder
        if (commitment_) {                            // the commitment guest's output is cheap enough to compute without proving
            r.output.assign(32, 0);
            zkhip_sha256_digest(input.cbor.data(), input.cbor.size(), r.output.data());
        }
        r.proof = {0, 0, 0, 0};
        r.ok = true;
        return r;
    }
    if (mode_ == ProverType::Network) throw std::runtime_error("network proving is not provided by the HIP backend");
    // Local and Hip both mean "prove on this machine"; there is no CPU path in libzkhip
    if (commitment_) {
        // the input-commitment guest: SHA-256 of the CBOR input through the chip, one proof, verified like sp1.rs:120
        if (devices_.empty()) throw std::runtime_error("device list is empty");
        const size_t padded = ((input.cbor.size() + 9 + 63) / 64) * 64;
        int log_n = 6;                                   // 64 rows per block, block count rounded up to a power of two
        while (((size_t)1 << (log_n - 6)) < padded / 64) log_n++;
        zkhip_params prm{1, plan_.num_queries, plan_.pow_bits, 0, 0, 0, 0, 0};
        if (backend_ == Backend::Risc0) {
            int lf = 8;
            while (lf > log_n || (log_n - lf) % 4 != 0) lf--;
            const bool defaults = plan_.num_queries == 100 && plan_.pow_bits == 16;
            prm = zkhip_params{2, defaults ? 50 : plan_.num_queries, defaults ? 0 : plan_.pow_bits, 0, 4, lf, 24, 0};
        }
        constexpr size_t ONE_PROOF = ((size_t)1 << 20) - 9;          // bytes whose padded form fits 2^14 blocks = one chip proof of 2^20 rows
        if (input.cbor.size() > ONE_PROOF && backend_ == Backend::Sp1) {
            // a large transcript (BASELINE configs[3]): SHA-256 as a CHAIN of shard proofs, dealt over the prover's devices; verified like sp1.rs:120
            if (!vk_.empty()) throw std::runtime_error("setup: the keyed machine takes inputs of up to 1 MiB");
            const int k = 14;
            const size_t n = zkhip_sha256_sharded_count(input.cbor.size(), k), stride = zkhip_sha256_shard_proof_size(k, &prm);
            if (n == 0 || stride == 0) throw std::runtime_error(std::string("input commitment: ") + zkhip_last_error());
            if (compress_) {
                // core -> compress (sp1.rs:116) on the chain: the shards (dealt over the devices) verified in-circuit on the first device -- ONE
                // proof.  Blob: the chaining values, the proof, the key of (shard count, shapes), the length; checked like sp1.rs:120 with
                // the key this context made (a consumer derives it on its host: verify_commitment_blob)
                const size_t ccap = zkhip_sha256_compressed_proof_size(input.cbor.size(), k, &prm, &prm);
                if (ccap == 0) throw std::runtime_error(std::string("input commitment, compressed: ") + zkhip_last_error());
                CtxGuard g;
                constexpr size_t PLACEHOLDER = 256;
                if (!g.take(devices_[0], PLACEHOLDER)) {
                    g.device = devices_[0]; g.trace_bytes = PLACEHOLDER;
                    if (zkhip_ctx_create(devices_[0], nullptr, &g.ctx) != ZKHIP_OK) fail_zkhip("zkhip_ctx_create");
                    if (zkhip_malloc(g.ctx, PLACEHOLDER, &g.d_trace) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
                }
                struct KeyGuard { zkhip_machine_key* k = nullptr; ~KeyGuard() { if (k) zkhip_machine_key_destroy(k); } } ck;
                uint32_t root[8];
                if (zkhip_sha256_compress_setup(g.ctx, input.cbor.size(), k, &prm, &prm, &ck.k, root) != ZKHIP_OK) fail_zkhip("zkhip_sha256_compress_setup");
                std::vector<uint32_t> chain((n + 1) * 8);
                std::vector<uint8_t> proof(ccap);
                uint8_t digest32[32];
                size_t len = 0;
                int reason = 0;
                if (zkhip_prove_sha256_compressed(g.ctx, ck.k, devices_.data(), (int)devices_.size(), input.cbor.data(), input.cbor.size(), k, &prm, &prm, 2, digest32, chain.data(),
                                                  proof.data(), ccap, &len) != ZKHIP_OK)
                    fail_zkhip("zkhip_prove_sha256_compressed");
                g.healthy = true;
                proof.resize(len);
                if (zkhip_verify_sha256_compressed(proof.data(), len, digest32, (uint64_t)input.cbor.size(), chain.data(), k, root, &prm, &prm, &reason) != ZKHIP_OK)
                    fail_zkhip("zkhip_verify_sha256_compressed");
                std::vector<std::vector<uint8_t>> entries;
                entries.emplace_back((const uint8_t*)chain.data(), (const uint8_t*)chain.data() + chain.size() * 4);
                entries.push_back(std::move(proof));
                entries.emplace_back((const uint8_t*)root, (const uint8_t*)root + 32);
                entries.push_back(length_entry(input.cbor.size()));
                r.output.assign(digest32, digest32 + 32);
                r.proof = pack_shard_proofs(entries, BATCH_FLAG_INPUT_SHA256 | BATCH_FLAG_CHAINED | BATCH_FLAG_COMPRESSED);
                r.ok = true;
                return r;
            }
            std::vector<uint8_t> buf(n * stride);
            std::vector<size_t> lens(n);
            std::vector<uint32_t> chain((n + 1) * 8);
            uint8_t digest32[32];
            if (zkhip_prove_sha256_sharded(devices_.data(), (int)devices_.size(), input.cbor.data(), input.cbor.size(), k, &prm, 2, digest32, chain.data(), buf.data(), stride, lens.data()) != ZKHIP_OK)
                fail_zkhip("zkhip_prove_sha256_sharded");
            size_t bad = 0;
            int reason = 0;
            if (zkhip_verify_sha256_sharded(buf.data(), stride, lens.data(), n, chain.data(), k, digest32, (uint64_t)input.cbor.size(), &prm, &bad, &reason) != ZKHIP_OK) fail_zkhip("zkhip_verify_sha256_sharded");
            std::vector<std::vector<uint8_t>> entries;
            entries.emplace_back((const uint8_t*)chain.data(), (const uint8_t*)chain.data() + chain.size() * 4);
            for (size_t s = 0; s < n; s++) entries.emplace_back(buf.begin() + (long)(s * stride), buf.begin() + (long)(s * stride + lens[s]));
            entries.push_back(length_entry(input.cbor.size()));
            r.output.assign(digest32, digest32 + 32);
            r.proof = pack_shard_proofs(entries, BATCH_FLAG_INPUT_SHA256 | BATCH_FLAG_CHAINED);
            r.ok = true;
            return r;
        }
        const bool keyed = !vk_.empty();
        if (keyed) {
            if (backend_ != Backend::Sp1) throw std::runtime_error("setup: the keyed machine uses the SP1 proof shape");
            const std::vector<uint32_t> pd = request_digest({}, elf);
            if (std::memcmp(vk_.data() + 32, pd.data(), 32) != 0) throw std::runtime_error("prove: the guest program is not the one setup() was called with");
        }
        const size_t cap = keyed ? zkhip_sha256_machine_proof_size(input.cbor.size(), &prm) : zkhip_sha256_proof_size(input.cbor.size(), &prm);
        if (cap == 0) throw std::runtime_error(std::string("input commitment: ") + zkhip_last_error());
        // a parked context when there is one (the chip keeps its trace in the context's own workspaces: a 256-byte placeholder
        // stands in for the trace buffer the pool is keyed by)
        CtxGuard g;
        constexpr size_t PLACEHOLDER = 256;
        if (!g.take(devices_[0], PLACEHOLDER)) {
            g.device = devices_[0]; g.trace_bytes = PLACEHOLDER;
            if (zkhip_ctx_create(devices_[0], nullptr, &g.ctx) != ZKHIP_OK) fail_zkhip("zkhip_ctx_create");
            if (zkhip_malloc(g.ctx, PLACEHOLDER, &g.d_trace) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
        }
        std::vector<uint8_t> proof(cap);
        size_t len = 0;
        uint8_t digest32[32];
        int reason = 0;
        if (keyed) {
            // pk: this context's key, made on first use (the commitment is a function of the proof shape alone: every context arrives
            // at the vk setup() returned, which is checked)
            if (!g.key && zkhip_sha256_setup(g.ctx, &prm, &g.key, g.key_root) != ZKHIP_OK) fail_zkhip("zkhip_sha256_setup");
            if (std::memcmp(g.key_root, vk_.data(), 32) != 0) throw std::runtime_error("prove: this context's key does not match the verifying key of setup()");
            if (zkhip_prove_sha256_machine(g.ctx, g.key, input.cbor.data(), input.cbor.size(), &prm, digest32, proof.data(), cap, &len) != ZKHIP_OK)
                fail_zkhip("zkhip_prove_sha256_machine");
            g.healthy = true;
            proof.resize(len);
            if (zkhip_verify_sha256_machine(proof.data(), proof.size(), digest32, (uint64_t)input.cbor.size(), (const uint32_t*)vk_.data(), &prm, &reason) != ZKHIP_OK)   // sp1.rs:120
                fail_zkhip("zkhip_verify_sha256_machine");
            r.vk = vk_;
        } else {
            if (zkhip_prove_sha256(g.ctx, input.cbor.data(), input.cbor.size(), &prm, digest32, proof.data(), cap, &len) != ZKHIP_OK) fail_zkhip("zkhip_prove_sha256");
            g.healthy = true;
            proof.resize(len);
            if (zkhip_verify_sha256(proof.data(), proof.size(), digest32, (uint64_t)input.cbor.size(), &prm, &reason) != ZKHIP_OK) fail_zkhip("zkhip_verify_sha256");
        }
        r.output.assign(digest32, digest32 + 32);
        r.proof = pack_shard_proofs({proof, length_entry(input.cbor.size())}, BATCH_FLAG_INPUT_SHA256 | (keyed ? BATCH_FLAG_KEYED : 0u));
        r.ok = true;
        return r;
    }
    if (machine_) {
        // synthetic shards in SP1's shard structure: every shard ONE keyed-machine proof (version 11) against ONE key, checked like sp1.rs:120
        if (backend_ != Backend::Sp1) throw std::runtime_error("machine shards use the SP1 proof shape");
        if (mplan_.shards == 0) throw std::runtime_error("shard plan is empty");
        if (devices_.empty()) throw std::runtime_error("device list is empty");
        const zkhip_params prm{1, mplan_.num_queries, mplan_.pow_bits, 0, 0, 0, 0, 0};
        if (!vk_.empty()) {
            const std::vector<uint32_t> pd = request_digest({}, elf);
            if (std::memcmp(vk_.data() + 32, pd.data(), 32) != 0) throw std::runtime_error("prove: the guest program is not the one setup() was called with");
        }
        MachineHolder shape;
        const uint32_t zero[8] = {0};
        shape.build_plan(mplan_, zero);
        const int nc = shape.desc.n_chips;
        const size_t cap = zkhip_machine_proof_size_keyed(shape.lns.data(), shape.widths.data(), shape.pres.data(), shape.pp.data(), shape.pw_.data(), shape.tp.data(), shape.tw_.data(), nc, &prm,
                                                          MACHINE_PUBLICS);
        if (cap == 0) throw std::runtime_error(std::string("bad machine plan: ") + zkhip_last_error());
        const uint64_t seed = stream_seed(digest);
        uint32_t in_flight = mplan_.in_flight;
        if (in_flight == 0) {
            const char* e = std::getenv("ZKTLS_HIP_IN_FLIGHT");
            in_flight = e && std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : 4u;
        }
        if (in_flight > mplan_.shards) in_flight = mplan_.shards;
        std::vector<size_t> offsets;
        const size_t bytes = plan_trace_bytes(mplan_, &offsets);
        std::vector<std::vector<uint8_t>> proofs(mplan_.shards);
        const int n_dev = (int)devices_.size();
        std::vector<std::atomic<uint32_t>> next(n_dev);
        for (auto& a : next) a.store(0);
        std::atomic<bool> failed{false};
        std::mutex err_mu;
        std::string first_error;
        uint32_t root[8] = {0};
        bool have_root = false;
        if (!vk_.empty()) { std::memcpy(root, vk_.data(), 32); have_root = true; }
        auto worker = [&](int slot) {
            try {
                const int device = devices_[slot];
                CtxGuard g;
                if (!g.take(device, bytes)) {
                    g.device = device; g.trace_bytes = bytes;
                    if (zkhip_ctx_create(device, nullptr, &g.ctx) != ZKHIP_OK) fail_zkhip("zkhip_ctx_create");
                    if (zkhip_malloc(g.ctx, bytes, &g.d_trace) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
                }
                // pk: a key belongs to the context it was made with; every context arrives at ONE root (setup()'s, when it was called)
                struct KeyGuard { zkhip_ctx* c; zkhip_machine_key* k = nullptr; ~KeyGuard() { if (k) { (void)zkhip_ctx_sync(c); zkhip_machine_key_destroy(k); } } } kg{g.ctx};
                uint32_t mine[8];
                machine_setup_on(g.ctx, mplan_, elf, prm, &kg.k, mine);
                {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!have_root) { std::memcpy(root, mine, 32); have_root = true; }
                    else if (std::memcmp(root, mine, 32) != 0) throw std::runtime_error("prove: this context's key does not match the verifying key");
                }
                std::vector<zkhip_chip> chips((size_t)nc);
                for (;;) {
                    const uint64_t s = (uint64_t)slot + (uint64_t)next[slot].fetch_add(1) * (uint64_t)n_dev;
                    if (s >= mplan_.shards || failed.load()) break;
                    if (zkhip_shard_device((int)s, devices_.data(), n_dev) != device) throw std::runtime_error("shard dealt to the wrong device");
                    for (int c = 0; c < nc; c++) {
                        const ChipPlan& ch = mplan_.chips[(size_t)c];
                        uint32_t* d = (uint32_t*)((uint8_t*)g.d_trace + offsets[(size_t)c]);
                        const int rc = ch.partner < 0
                            ? zkhip_gen_trace_logup(g.ctx, seed, 100u * s + (uint64_t)c, ch.log_n, ch.width, (int)ch.pairs, d, ch.width)
                            : zkhip_gen_trace_logup_cross(g.ctx, seed, 100u * s + (uint64_t)c, 100u * s + (uint64_t)ch.partner, ch.log_n, ch.width, mplan_.chips[(size_t)ch.partner].width,
                                                          (int)ch.pairs, d, ch.width);
                        if (rc != ZKHIP_OK) fail_zkhip("zkhip_gen_trace_logup");
                        chips[(size_t)c] = zkhip_chip{d + ch.pre_width, ch.width, ch.log_n, ch.width - ch.pre_width, 0, -1};      // the main columns: a view of [preprocessed | main]
                    }
                    std::vector<uint32_t> pv(digest);
                    pv.push_back((uint32_t)s);
                    std::vector<uint8_t> proof(cap);
                    size_t len = 0;
                    if (zkhip_prove_machine_keyed(g.ctx, kg.k, chips.data(), shape.pp.data(), shape.pw_.data(), shape.tp.data(), shape.tw_.data(), nc, pv.data(), pv.size(), &prm, proof.data(), cap,
                                                  &len) != ZKHIP_OK)
                        fail_zkhip("zkhip_prove_machine_keyed");
                    proof.resize(len);
                    int reason = 0;
                    if (zkhip_verify_machine_keyed(proof.data(), len, shape.lns.data(), shape.widths.data(), shape.pres.data(), mine, shape.pp.data(), shape.pw_.data(), shape.tp.data(),
                                                   shape.tw_.data(), nc, pv.data(), pv.size(), &prm, &reason) != ZKHIP_OK)
                        fail_zkhip("zkhip_verify_machine_keyed");          // sp1.rs:120: the prover checks its own proof
                    proofs[s] = std::move(proof);
                }
                g.healthy = true;
            } catch (const std::exception& e) {
                failed.store(true);
                std::lock_guard<std::mutex> lk(err_mu);
                if (first_error.empty()) first_error = e.what();
            } catch (...) {
                failed.store(true);
                std::lock_guard<std::mutex> lk(err_mu);
                if (first_error.empty()) first_error = "unknown failure in a shard worker";
            }
        };
        if (in_flight <= 1 && n_dev == 1) {
            worker(0);
        } else {
            std::vector<std::thread> pool;
            for (int slot = 0; slot < n_dev; slot++) {
                const uint32_t mine = (mplan_.shards - (uint32_t)slot + (uint32_t)n_dev - 1) / (uint32_t)n_dev;
                for (uint32_t t = 0; t < (mine < in_flight ? mine : in_flight); t++) pool.emplace_back(worker, slot);
            }
            for (auto& t : pool) t.join();
        }
        if (failed.load()) throw std::runtime_error(first_error);
        r.vk.assign(64, 0);
        std::memcpy(r.vk.data(), root, 32);
        const std::vector<uint32_t> pd = request_digest({}, elf);
        std::memcpy(r.vk.data() + 32, pd.data(), 32);
        r.proof = compress_ ? compress_machine_proofs(devices_[0], mplan_, root, digest, proofs) : pack_shard_proofs(proofs, BATCH_FLAG_SYNTHETIC | BATCH_FLAG_MACHINE);
        r.ok = true;
        return r;
    }
    if (!synthetic_)
        throw std::runtime_error("no shard source: the zkVM executor that turns (input, ELF) into shard traces is not part of the HIP backend; "
                                 "with_synthetic(plan) opts into proving synthetic shards (the blob is then flagged SYNTHETIC), with_input_commitment() into the SHA-256-of-the-input guest");
    if (plan_.shards == 0) throw std::runtime_error("shard plan is empty");
    if (devices_.empty()) throw std::runtime_error("device list is empty");
    zkhip_params prm{1, plan_.num_queries, plan_.pow_bits, 0, 0, 0, 0, 0};
    if (backend_ == Backend::Risc0) {
        // RISC Zero's shape; the final polynomial shrinks for segments too small for 256 coefficients
        int lf = 8;
        while (lf > plan_.log_n || (plan_.log_n - lf) % 4 != 0) lf--;
        const bool defaults = plan_.num_queries == 100 && plan_.pow_bits == 16;
        prm = zkhip_params{2, defaults ? 50 : plan_.num_queries, defaults ? 0 : plan_.pow_bits, 0, 4, lf, 24, 0};
    }
    const size_t cap = zkhip_proof_size(plan_.log_n, plan_.width, &prm, 9);
    if (cap == 0) throw std::runtime_error(std::string("bad shard plan: ") + zkhip_last_error());
    uint64_t seed = 0;
    for (int i = 0; i < 4; i++) seed = (seed << 16) ^ digest[i];
    // Shards are independent: `in_flight` of them are proven at the same time, each on its own context (= HIP stream) and
    // host thread, so the latency-bound stretches of one proof (small Merkle levels, transcript round trips, the CPU
    // verification of the finished proof) hide under the throughput-bound kernels of the others.
    uint32_t in_flight = plan_.in_flight;
    if (in_flight == 0) {
        const char* e = std::getenv("ZKTLS_HIP_IN_FLIGHT");
        in_flight = e && std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : 4u;
    }
    if (in_flight > plan_.shards) in_flight = plan_.shards;
    const size_t words = ((size_t)1 << plan_.log_n) * plan_.width;
    std::vector<std::vector<uint8_t>> proofs(plan_.shards);
    const int n_dev = (int)devices_.size();
    std::vector<std::atomic<uint32_t>> next(n_dev);          // per device: how many of its shards were handed out
    for (auto& a : next) a.store(0);
    std::atomic<bool> failed{false};
    std::mutex err_mu;
    std::string first_error;
    auto worker = [&](int slot) {
        try {
            const int device = devices_[slot];
            CtxGuard g;
            if (!g.take(device, words * 4)) {
                g.device = device; g.trace_bytes = words * 4;
                if (zkhip_ctx_create(device, nullptr, &g.ctx) != ZKHIP_OK) fail_zkhip("zkhip_ctx_create");
                if (zkhip_malloc(g.ctx, words * 4, &g.d_trace) != ZKHIP_OK) fail_zkhip("zkhip_malloc");
            }
            for (;;) {
                // the k-th shard of this device: shard s runs on devices[s mod n], as zkhip_shard_device deals them
                const uint64_t s = (uint64_t)slot + (uint64_t)next[slot].fetch_add(1) * (uint64_t)n_dev;
                if (s >= plan_.shards || failed.load()) break;
                if (zkhip_shard_device((int)s, devices_.data(), n_dev) != device) throw std::runtime_error("shard dealt to the wrong device");
                std::vector<uint32_t> pv(digest);
                pv.push_back((uint32_t)s);
                if (zkhip_gen_trace(g.ctx, seed, s, plan_.log_n, plan_.width, (uint32_t*)g.d_trace, plan_.width) != ZKHIP_OK)
                    fail_zkhip("zkhip_gen_trace");
                std::vector<uint8_t> proof(cap);
                size_t len = 0;
                if (zkhip_prove_shard(g.ctx, (const uint32_t*)g.d_trace, plan_.width, plan_.log_n, plan_.width, pv.data(), pv.size(),
                                      &prm, proof.data(), cap, &len) != ZKHIP_OK)
                    fail_zkhip("zkhip_prove_shard");
                proof.resize(len);
                int reason = 0;
                if (zkhip_verify_shard(proof.data(), proof.size(), plan_.log_n, plan_.width, pv.data(), pv.size(), &prm, &reason) != ZKHIP_OK)
                    fail_zkhip("zkhip_verify_shard");   // sp1.rs:120: the prover checks its own proof
                proofs[s] = std::move(proof);
            }
            g.healthy = true;
        } catch (const std::exception& e) {              // a panic in one worker must not escape its thread (sp1.rs:85)
            failed.store(true);
            std::lock_guard<std::mutex> lk(err_mu);
            if (first_error.empty()) first_error = e.what();
        } catch (...) {
            failed.store(true);
            std::lock_guard<std::mutex> lk(err_mu);
            if (first_error.empty()) first_error = "unknown failure in a shard worker";
        }
    };
    if (in_flight <= 1 && n_dev == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int slot = 0; slot < n_dev; slot++) {
            const uint32_t mine = (plan_.shards - (uint32_t)slot + (uint32_t)n_dev - 1) / (uint32_t)n_dev;
            for (uint32_t t = 0; t < (mine < in_flight ? mine : in_flight); t++) pool.emplace_back(worker, slot);
        }
        for (auto& t : pool) t.join();
    }
    if (failed.load()) throw std::runtime_error(first_error);
    if (compress_) {
        if (backend_ != Backend::Sp1) throw std::runtime_error("with_compress: the shard verifier takes SP1-shape shard proofs");
        r.proof = compress_shard_proofs(devices_, plan_, prm, digest, proofs);
        r.ok = true;
        return r;
    }
    r.proof = pack_shard_proofs(proofs, BATCH_FLAG_SYNTHETIC);
    r.ok = true;
    return r;
}

void set_compress_join_size(uint32_t shard_proofs_per_join) { g_join_size.store(shard_proofs_per_join); }

uint32_t compress_join_size(const ShardPlan& plan) {
    const uint32_t shards = plan.shards ? plan.shards : 1u;
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};      // (the outer proof's shape: blowup 2, the plan's queries and proof-of-work bits)
    uint32_t most = (uint32_t)zkhip_shard_verifier_max_proofs(plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, &outer);
    if (most == 0) most = 1;                           // (a shape the machine refuses: the join itself will say so)
    const uint32_t cap = g_join_size.load();           // (set_compress_join_size: smaller joins -- bounded latency per join, more of them, a tree above)
    if (cap && cap < most) most = cap;
    if (shards <= most) return shards;
    const uint32_t joins = (shards + most - 1u) / most;
    return (shards + joins - 1u) / joins;
}

int verify_compressed_blob(const std::vector<uint8_t>& blob, const ShardPlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf,
                           const uint32_t key[8], int* reason) {
    std::vector<std::vector<uint8_t>> entries;
    uint32_t flags = 0;
    if (!unpack_shard_proofs(blob, &entries, &flags) || entries.size() < 2 || !(flags & BATCH_FLAG_COMPRESSED) || entries.back().size() != 36) return -1;
    uint32_t cnt = 0;
    std::memcpy(&cnt, entries.back().data() + 32, 4);
    if (cnt != plan.shards || cnt == 0 || std::memcmp(entries.back().data(), key, 32) != 0) return -1;        // (the key is the CALLER's: the blob's copy is informative)
    const uint32_t J = compress_join_size(plan), n_joins = (cnt + J - 1) / J;
    const std::vector<uint32_t> digest = request_digest(cbor, elf);
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    if (flags & BATCH_FLAG_TREE) {
        // ONE proof that verifies the joins: its key is derived here, on the host, from the join machine (a function of the plan) and the join key
        if (n_joins < 2 || entries.size() != 2) return -1;
        JoinMachine jm;
        if (!jm.build(plan, J, key)) return -1;
        uint32_t tvk[8];
        if (!machine_verifier_key_host_cached(jm.desc, n_joins, outer, tvk)) return -1;      // (derived once per plan and join key: hundreds of milliseconds at the bench's shape)
        std::vector<uint32_t> pvs;
        for (uint32_t c = 0; c < n_joins; c++)
            for (uint32_t k = 0; k < J; k++) {
                const uint32_t sidx = c * J + k < cnt ? c * J + k : cnt - 1;
                pvs.insert(pvs.end(), digest.begin(), digest.end());
                pvs.push_back(sidx);
            }
        int why = 0;
        const int rc = zkhip_verify_machine_recursive(&jm.desc, entries[0].data(), entries[0].size(), pvs.data(), 9u * J, n_joins, tvk, &outer, &why);
        if (reason) *reason = why;
        return rc == ZKHIP_OK ? 0 : -2;
    }
    if (entries.size() != (size_t)n_joins + 1) return -1;
    for (uint32_t c = 0; c < n_joins; c++) {
        std::vector<uint32_t> pvs;
        for (uint32_t k = 0; k < J; k++) {
            const uint32_t sidx = c * J + k < cnt ? c * J + k : cnt - 1;      // the last join repeats the last shard (one shape, one key)
            pvs.insert(pvs.end(), digest.begin(), digest.end());
            pvs.push_back(sidx);
        }
        int why = 0;
        const int rc = zkhip_verify_shard_recursive(entries[c].data(), entries[c].size(), plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, pvs.data(), 9, J, key, &outer, &why);
        if (reason) *reason = why;
        if (rc != ZKHIP_OK) return -2;
    }
    return 0;
}

bool compress_key(int device, const ShardPlan& plan, uint32_t key_out[8], std::string* error) {
    zkhip_ctx* ctx = nullptr;
    zkhip_machine_key* key = nullptr;
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    bool ok = zkhip_ctx_create(device, nullptr, &ctx) == ZKHIP_OK &&
              zkhip_shard_verifier_setup(ctx, plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, compress_join_size(plan), &outer, &key, key_out) == ZKHIP_OK;
    if (!ok && error) *error = zkhip_last_error();
    if (key) { (void)zkhip_ctx_sync(ctx); zkhip_machine_key_destroy(key); }
    if (ctx) zkhip_ctx_destroy(ctx);
    return ok;
}

// the same key on the host's cores: no context, no device (zkhip_shard_verifier_key_host) -- what a verifier without a GPU calls
bool compress_key_host(const ShardPlan& plan, uint32_t key_out[8], std::string* error) {
    const zkhip_params outer{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
    const bool ok = zkhip_shard_verifier_key_host(plan.log_n, plan.width, (size_t)plan.num_queries, plan.pow_bits, 9, compress_join_size(plan), &outer, key_out) == ZKHIP_OK;
    if (!ok && error) *error = zkhip_last_error();
    return ok;
}

// core -> compress as a step of its own (sp1-cuda's prove_core / compress pair behind sp1.rs:116): the SYNTHETIC batch blob prove() returned for
// (plan, input, ELF) -> the COMPRESSED blob with_compress() would have returned.  Every shard proof is verified on the way: the machine's tables
// ARE the verification, a proof that does not verify has no witness.
ProveResult compress_blob(int device, const ShardPlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf, const std::vector<uint8_t>& blob) {
    ProveResult r;
    try {
        std::vector<std::vector<uint8_t>> proofs;
        uint32_t flags = 0;
        if (!unpack_shard_proofs(blob, &proofs, &flags)) throw std::runtime_error("compress: not a batch blob");
        if (flags != BATCH_FLAG_SYNTHETIC) throw std::runtime_error("compress: the blob must hold SP1-shape shard proofs of a synthetic plan (flags = SYNTHETIC)");
        if (plan.shards == 0 || proofs.size() != plan.shards) throw std::runtime_error("compress: the blob does not hold the plan's number of shard proofs");
        const zkhip_params prm{1, plan.num_queries, plan.pow_bits, 0, 0, 0, 0, 0};
        const std::vector<uint32_t> digest = request_digest(cbor, elf);
        r.output.resize(32);
        std::memcpy(r.output.data(), digest.data(), 32);
        r.proof = compress_shard_proofs(std::vector<int>{device}, plan, prm, digest, proofs);
        r.ok = true;
    } catch (const std::exception& e) {
        r = ProveResult{};
        r.error = e.what();
    } catch (...) {
        r = ProveResult{};
        r.error = "unknown failure in compress_blob";
    }
    return r;
}

void release_cached() {
    std::vector<Slot> all;
    { std::lock_guard<std::mutex> lk(g_slots_mu); all.swap(g_slots); }
    std::vector<JoinKey> keys;
    { std::lock_guard<std::mutex> lk(g_slots_mu); keys.swap(g_join_keys); }
    for (auto& k : keys) { zkhip_machine_key_destroy(k.key); zkhip_ctx_destroy(k.ctx); }
    std::vector<MachineJoinKey> mkeys;
    { std::lock_guard<std::mutex> lk(g_slots_mu); mkeys.swap(g_machine_join_keys); }
    for (auto& k : mkeys) { zkhip_machine_key_destroy(k.key); zkhip_ctx_destroy(k.ctx); }
    { std::lock_guard<std::mutex> lk(g_derived_mu); g_derived.clear(); }
    for (auto& e : all) { if (e.key) zkhip_machine_key_destroy(e.key); zkhip_free(e.ctx, e.d_trace); zkhip_ctx_destroy(e.ctx); }
}

}  // namespace zktls

// ---- flat C surface so the tests (ctypes) and other FFIs can drive the mirror ----
extern "C" {

// frees the contexts and trace buffers parked by finished workers
void zktls_release_cached(void) { zktls::release_cached(); }
// shard proofs per join of the compress stage (0: as many as one join holds -- 136 of the headline shape); with fewer, an execution takes several
// joins and ONE more proof above them (the tree: blob flag TREE).  Process-wide; prover and verifier must agree (it is part of the plan)
void zktls_set_compress_join_size(uint32_t shard_proofs_per_join) { zktls::set_compress_join_size(shard_proofs_per_join); }

struct zktls_shard_plan { int32_t log_n; uint32_t width; uint32_t shards; int32_t num_queries; int32_t pow_bits; };

// mode: 0 mock, 1 local, 2 hip, 3 network.  Returns 0 on success; on failure copies the
// message into err.  *output / *proof are malloc'd (zktls_free).  A non-null `plan` IS the explicit opt-in to synthetic
// shards (HipGuestProver::with_synthetic); with plan == NULL modes 1 / 2 fail with "no shard source".  `device` >= 0: that
// device; device < 0: every visible device, shards dealt round-robin.
static int guest_prove(zktls::Backend backend, int device, int mode, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len,
                       const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                       size_t* proof_len, char* err, size_t err_cap, bool compress = false) {
    zktls::HipGuestProver p(device < 0 ? 0 : device, backend);
    if (compress) p.with_compress();
    if (device < 0) {
        std::vector<int> all;
        for (int d = 0; d < zkhip_device_count(); d++) all.push_back(d);
        if (!all.empty()) p.with_devices(all);
    }
    switch (mode) {
        case 0: p.mock(); break;
        case 1: p.local(); break;
        case 2: p.hip(); break;
        default: p.network(); break;
    }
    if (plan) {
        zktls::ShardPlan sp;
        sp.log_n = plan->log_n; sp.width = plan->width; sp.shards = plan->shards;
        sp.num_queries = plan->num_queries; sp.pow_bits = plan->pow_bits;
        p.with_synthetic(sp);
    }
    zktls::GuestInput in;
    in.cbor.assign(cbor, cbor + cbor_len);
    std::vector<uint8_t> program(elf, elf + elf_len);
    zktls::ProveResult r = p.prove(in, program);
    if (!r.ok) {
        if (err && err_cap) { std::strncpy(err, r.error.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        return -1;
    }
    *output_len = r.output.size();
    *output = (uint8_t*)std::malloc(r.output.size() ? r.output.size() : 1);
    std::memcpy(*output, r.output.data(), r.output.size());
    *proof_len = r.proof.size();
    *proof = (uint8_t*)std::malloc(r.proof.size() ? r.proof.size() : 1);
    std::memcpy(*proof, r.proof.data(), r.proof.size());
    return 0;
}
// compress as its own step: the batch blob of zktls_guest_prove (mode 2, the same plan / input / ELF) -> the compressed blob (malloc'd, zktls_free)
int zktls_compress_blob(int device, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len,
                        const uint8_t* blob, size_t blob_len, uint8_t** proof, size_t* proof_len, char* err, size_t err_cap) {
    if (!plan || !blob || !proof || !proof_len) { if (err && err_cap) { std::strncpy(err, "compress: null argument", err_cap - 1); err[err_cap - 1] = 0; } return -1; }
    zktls::ShardPlan sp;
    sp.log_n = plan->log_n; sp.width = plan->width; sp.shards = plan->shards; sp.num_queries = plan->num_queries; sp.pow_bits = plan->pow_bits;
    const zktls::ProveResult r = zktls::compress_blob(device < 0 ? 0 : device, sp, std::vector<uint8_t>(cbor, cbor + cbor_len), std::vector<uint8_t>(elf, elf + elf_len),
                                                      std::vector<uint8_t>(blob, blob + blob_len));
    if (!r.ok) { if (err && err_cap) { std::strncpy(err, r.error.c_str(), err_cap - 1); err[err_cap - 1] = 0; } return -1; }
    *proof_len = r.proof.size();
    *proof = (uint8_t*)std::malloc(r.proof.size() ? r.proof.size() : 1);
    std::memcpy(*proof, r.proof.data(), r.proof.size());
    return 0;
}
// the same with the compress stage behind it (HipGuestProver::with_compress): the blob carries ONE proof that verifies the shards
int zktls_guest_prove_compressed(int device, int mode, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len,
                                 uint8_t** output, size_t* output_len, uint8_t** proof, size_t* proof_len, char* err, size_t err_cap) {
    if (!plan) { if (err && err_cap) { std::strncpy(err, "compress needs a shard plan", err_cap - 1); err[err_cap - 1] = 0; } return -1; }
    return guest_prove(zktls::Backend::Sp1, device, mode, plan, cbor, cbor_len, elf, elf_len, output, output_len, proof, proof_len, err, err_cap, true);
}
// the key of the plan's shape (device work, once per shape), and the host-only check of a COMPRESSED blob against it: 0 / -1 / -2 (verify_compressed_blob)
int zktls_compress_key(int device, const zktls_shard_plan* plan, uint32_t key[8], char* err, size_t err_cap) {
    zktls::ShardPlan sp;
    sp.log_n = plan->log_n; sp.width = plan->width; sp.shards = plan->shards; sp.num_queries = plan->num_queries; sp.pow_bits = plan->pow_bits;
    std::string e;
    if (zktls::compress_key(device, sp, key, &e)) return 0;
    if (err && err_cap) { std::strncpy(err, e.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
}
// ... and without a device: the verifier's side of `client.verify` (sp1.rs:120) for a compressed blob needs this and zktls_verify_compressed_blob only
int zktls_compress_key_host(const zktls_shard_plan* plan, uint32_t key[8], char* err, size_t err_cap) {
    zktls::ShardPlan sp;
    sp.log_n = plan->log_n; sp.width = plan->width; sp.shards = plan->shards; sp.num_queries = plan->num_queries; sp.pow_bits = plan->pow_bits;
    std::string e;
    if (zktls::compress_key_host(sp, key, &e)) return 0;
    if (err && err_cap) { std::strncpy(err, e.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
}
int zktls_verify_compressed_blob(const uint8_t* blob, size_t len, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len,
                                 const uint32_t key[8], int* reason) {
    zktls::ShardPlan sp;
    sp.log_n = plan->log_n; sp.width = plan->width; sp.shards = plan->shards; sp.num_queries = plan->num_queries; sp.pow_bits = plan->pow_bits;
    return zktls::verify_compressed_blob(std::vector<uint8_t>(blob, blob + len), sp, std::vector<uint8_t>(cbor, cbor + cbor_len), std::vector<uint8_t>(elf, elf + elf_len), key, reason);
}
int zktls_guest_prove(int device, int mode, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len,
                      const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                      size_t* proof_len, char* err, size_t err_cap) {
    return guest_prove(zktls::Backend::Sp1, device, mode, plan, cbor, cbor_len, elf, elf_len, output, output_len, proof, proof_len, err, err_cap);
}
// the `-p r0` twin (crates/guest-prover-r0): same arguments, RISC0_* environment, RISC-Zero-like segment proofs
int zktls_guest_prove_r0(int device, int mode, const zktls_shard_plan* plan, const uint8_t* cbor, size_t cbor_len,
                         const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                         size_t* proof_len, char* err, size_t err_cap) {
    return guest_prove(zktls::Backend::Risc0, device, mode, plan, cbor, cbor_len, elf, elf_len, output, output_len, proof, proof_len, err, err_cap);
}
// the input-commitment guest (HipGuestProver::with_input_commitment): output = SHA-256 of the CBOR input, proof = a batch blob
// flagged INPUT_SHA256 holding one SHA-256 chip proof; backend 0 SP1 shape, 1 RISC Zero shape
static int prove_commitment(bool compress, int backend, int device, int mode, int num_queries, int pow_bits, const uint8_t* cbor, size_t cbor_len,
                            const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                            size_t* proof_len, char* err, size_t err_cap) {
    zktls::HipGuestProver p(device < 0 ? 0 : device, backend ? zktls::Backend::Risc0 : zktls::Backend::Sp1);
    if (compress) p.with_compress();
    switch (mode) {
        case 0: p.mock(); break;
        case 1: p.local(); break;
        case 2: p.hip(); break;
        default: p.network(); break;
    }
    zktls::ShardPlan sp;
    sp.num_queries = num_queries; sp.pow_bits = pow_bits;
    p.with_input_commitment(sp);
    zktls::GuestInput in;
    in.cbor.assign(cbor, cbor + cbor_len);
    zktls::ProveResult r = p.prove(in, std::vector<uint8_t>(elf, elf + elf_len));
    if (!r.ok) {
        if (err && err_cap) { std::strncpy(err, r.error.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        return -1;
    }
    *output_len = r.output.size();
    *output = (uint8_t*)std::malloc(r.output.size() ? r.output.size() : 1);
    std::memcpy(*output, r.output.data(), r.output.size());
    *proof_len = r.proof.size();
    *proof = (uint8_t*)std::malloc(r.proof.size() ? r.proof.size() : 1);
    std::memcpy(*proof, r.proof.data(), r.proof.size());
    return 0;
}
int zktls_guest_prove_commitment(int backend, int device, int mode, int num_queries, int pow_bits, const uint8_t* cbor, size_t cbor_len,
                                 const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                                 size_t* proof_len, char* err, size_t err_cap) {
    return prove_commitment(false, backend, device, mode, num_queries, pow_bits, cbor, cbor_len, elf, elf_len, output, output_len, proof, proof_len, err, err_cap);
}
// ... with the COMPRESS stage (sp1.rs:116: core -> compress): an input beyond one chip proof leaves as ONE proof instead of a chain of shard
// proofs (blob flags INPUT_SHA256 | CHAINED | COMPRESSED); smaller inputs are one proof already and come out as above
int zktls_guest_prove_commitment_compressed(int device, int mode, int num_queries, int pow_bits, const uint8_t* cbor, size_t cbor_len,
                                            const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                                            size_t* proof_len, char* err, size_t err_cap) {
    return prove_commitment(true, 0, device, mode, num_queries, pow_bits, cbor, cbor_len, elf, elf_len, output, output_len, proof, proof_len, err, err_cap);
}
// setup -> prove -> verify as the reference calls them (sp1.rs:113, :116, :120) on the input-commitment guest, SP1 backend: vk_out
// receives the 64-byte verifying key, the blob is flagged INPUT_SHA256 | KEYED
int zktls_guest_prove_commitment_keyed(int device, int mode, int num_queries, int pow_bits, const uint8_t* cbor, size_t cbor_len,
                                       const uint8_t* elf, size_t elf_len, uint8_t** output, size_t* output_len, uint8_t** proof,
                                       size_t* proof_len, uint8_t vk_out[64], char* err, size_t err_cap) {
    zktls::HipGuestProver p(device < 0 ? 0 : device, zktls::Backend::Sp1);
    switch (mode) {
        case 0: p.mock(); break;
        case 1: p.local(); break;
        case 2: p.hip(); break;
        default: p.network(); break;
    }
    zktls::ShardPlan sp;
    sp.num_queries = num_queries; sp.pow_bits = pow_bits;
    p.with_input_commitment(sp);
    const std::vector<uint8_t> program(elf, elf + elf_len);
    auto report = [&](const std::string& e) { if (err && err_cap) { std::strncpy(err, e.c_str(), err_cap - 1); err[err_cap - 1] = 0; } return -1; };
    const zktls::SetupResult s = p.setup(program);
    if (!s.ok) return report(s.error);
    zktls::GuestInput in;
    in.cbor.assign(cbor, cbor + cbor_len);
    zktls::ProveResult r = p.prove(in, program);
    if (!r.ok) return report(r.error);
    if (vk_out) std::memcpy(vk_out, s.vk.data(), 64);
    *output_len = r.output.size();
    *output = (uint8_t*)std::malloc(r.output.size() ? r.output.size() : 1);
    std::memcpy(*output, r.output.data(), r.output.size());
    *proof_len = r.proof.size();
    *proof = (uint8_t*)std::malloc(r.proof.size() ? r.proof.size() : 1);
    std::memcpy(*proof, r.proof.data(), r.proof.size());
    return 0;
}
int zktls_verify_commitment_blob(const uint8_t* blob, size_t len, const uint8_t output[32], const uint8_t* vk, size_t vk_len, int num_queries,
                                 int pow_bits, int* reason) {
    if (!blob || !output) return -1;
    return zktls::verify_commitment_blob(std::vector<uint8_t>(blob, blob + len), std::vector<uint8_t>(output, output + 32),
                                         vk ? std::vector<uint8_t>(vk, vk + vk_len) : std::vector<uint8_t>(), num_queries, pow_bits, reason);
}
// the same check for either backend's proof shape (0: SP1, 1: RISC Zero), as zktls_guest_prove_commitment takes it
// the length (bytes) of the input a commitment blob speaks about -- the other half of its statement "output = SHA-256 of a message of this length";
// a consumer that holds the input compares.  0 / -1 (not a commitment blob)
int zktls_commitment_blob_length(const uint8_t* blob, size_t len, uint64_t* message_len) {
    return zktls::commitment_blob_length(std::vector<uint8_t>(blob, blob + len), message_len) ? 0 : -1;
}
int zktls_verify_commitment_blob_for(int backend, const uint8_t* blob, size_t len, const uint8_t output[32], const uint8_t* vk, size_t vk_len,
                                     int num_queries, int pow_bits, int* reason) {
    if (!blob || !output || backend < 0 || backend > 1) return -1;
    return zktls::verify_commitment_blob(std::vector<uint8_t>(blob, blob + len), std::vector<uint8_t>(output, output + 32),
                                         vk ? std::vector<uint8_t>(vk, vk + vk_len) : std::vector<uint8_t>(), num_queries, pow_bits, reason,
                                         backend == 1 ? zktls::Backend::Risc0 : zktls::Backend::Sp1);
}
// ---- machine shards (MachinePlan): SP1's shard structure through the same plug point ----
struct zktls_machine_plan {
    int32_t n_chips; const int32_t* log_ns; const uint32_t* widths; const uint32_t* pairs; const int32_t* partners; const uint32_t* pre_widths;      // per chip, tallest first
    uint32_t shards; int32_t num_queries; int32_t pow_bits; uint32_t in_flight;
};
static bool machine_plan_of(const zktls_machine_plan* plan, zktls::MachinePlan* out) {
    if (!plan || plan->n_chips <= 0 || plan->n_chips > 16 || !plan->log_ns || !plan->widths) return false;
    for (int c = 0; c < plan->n_chips; c++)
        out->chips.push_back(zktls::ChipPlan{plan->log_ns[c], plan->widths[c], plan->pairs ? plan->pairs[c] : 0u, plan->partners ? plan->partners[c] : -1, plan->pre_widths ? plan->pre_widths[c] : 0u});
    out->shards = plan->shards; out->num_queries = plan->num_queries; out->pow_bits = plan->pow_bits; out->in_flight = plan->in_flight;
    return true;
}
static void put_err(char* err, size_t err_cap, const std::string& what) {
    if (err && err_cap) { std::strncpy(err, what.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
}
// `let (pk, vk) = client.setup(elf)` (sp1.rs:113) for a machine plan: vk[64] = the key's root (8 LE words) + the program's digest
int zktls_machine_setup(int device, int mode, const zktls_machine_plan* plan, const uint8_t* elf, size_t elf_len, uint8_t vk[64], char* err, size_t err_cap) {
    zktls::MachinePlan mp;
    if (!machine_plan_of(plan, &mp) || !vk) { put_err(err, err_cap, "machine setup: bad plan"); return -1; }
    zktls::HipGuestProver p(device < 0 ? 0 : device);
    switch (mode) { case 0: p.mock(); break; case 1: p.local(); break; case 2: p.hip(); break; default: p.network(); break; }
    p.with_synthetic_machine(mp);
    const zktls::SetupResult s = p.setup(std::vector<uint8_t>(elf, elf + elf_len));
    if (!s.ok) { put_err(err, err_cap, s.error); return -1; }
    std::memcpy(vk, s.vk.data(), 64);
    return 0;
}
// ONE call of ZkProver::prove for an execution of machine shards; setup_first != 0: setup(elf) before it, as prove<P> does (sp1.rs:113-116); compress != 0: the
// compress stage behind it.  vk_out[64] receives the verifying key the proofs were checked against.  `device` < 0: every visible device.
int zktls_guest_prove_machine(int device, int mode, const zktls_machine_plan* plan, int compress, int setup_first, const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len,
                              uint8_t** output, size_t* output_len, uint8_t** proof, size_t* proof_len, uint8_t vk_out[64], char* err, size_t err_cap) {
    zktls::MachinePlan mp;
    if (!machine_plan_of(plan, &mp) || !output || !output_len || !proof || !proof_len) { put_err(err, err_cap, "machine prove: bad plan or null argument"); return -1; }
    zktls::HipGuestProver p(device < 0 ? 0 : device);
    if (device < 0) {
        std::vector<int> all;
        for (int d = 0; d < zkhip_device_count(); d++) all.push_back(d);
        if (!all.empty()) p.with_devices(all);
    }
    switch (mode) { case 0: p.mock(); break; case 1: p.local(); break; case 2: p.hip(); break; default: p.network(); break; }
    p.with_synthetic_machine(mp);
    if (compress) p.with_compress();
    const std::vector<uint8_t> program(elf, elf + elf_len);
    if (setup_first) {
        const zktls::SetupResult s = p.setup(program);
        if (!s.ok) { put_err(err, err_cap, s.error); return -1; }
    }
    zktls::GuestInput in;
    in.cbor.assign(cbor, cbor + cbor_len);
    const zktls::ProveResult r = p.prove(in, program);
    if (!r.ok) { put_err(err, err_cap, r.error); return -1; }
    *output_len = r.output.size();
    *output = (uint8_t*)std::malloc(r.output.size() ? r.output.size() : 1);
    std::memcpy(*output, r.output.data(), r.output.size());
    *proof_len = r.proof.size();
    *proof = (uint8_t*)std::malloc(r.proof.size() ? r.proof.size() : 1);
    std::memcpy(*proof, r.proof.data(), r.proof.size());
    if (vk_out) { std::memset(vk_out, 0, 64); if (r.vk.size() == 64) std::memcpy(vk_out, r.vk.data(), 64); }
    return 0;
}
// the host-only check of a machine blob (plain, COMPRESSED or COMPRESSED | TREE) against (plan, input, ELF, vk): 0 / -1 / -2 (verify_machine_blob)
int zktls_verify_machine_blob(const uint8_t* blob, size_t len, const zktls_machine_plan* plan, const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len, const uint8_t vk[64],
                              int* reason) {
    zktls::MachinePlan mp;
    if (!machine_plan_of(plan, &mp) || !blob || !vk) return -1;
    return zktls::verify_machine_blob(std::vector<uint8_t>(blob, blob + len), mp, std::vector<uint8_t>(cbor, cbor + cbor_len), std::vector<uint8_t>(elf, elf + elf_len),
                                      std::vector<uint8_t>(vk, vk + 64), reason);
}
// shard proofs per machine-mode join for the plan (0: a plan the library refuses)
uint32_t zktls_machine_join_size(const zktls_machine_plan* plan) {
    zktls::MachinePlan mp;
    if (!machine_plan_of(plan, &mp)) return 0;
    try { return zktls::machine_join_size(mp); } catch (...) { return 0; }
}
void zktls_free(void* p) { std::free(p); }
const char* zktls_current_risc0_prover_env(void) { const char* e = getenv("RISC0_PROVER"); return e ? e : ""; }
const char* zktls_current_risc0_dev_mode_env(void) { const char* e = getenv("RISC0_DEV_MODE"); return e ? e : ""; }
const char* zktls_current_sp1_prover_env(void) { const char* e = getenv("SP1_PROVER"); return e ? e : ""; }
int zktls_request_digest(const uint8_t* cbor, size_t cbor_len, const uint8_t* elf, size_t elf_len, uint32_t out[8]) {
    return zkhip_request_digest(cbor, cbor_len, elf, elf_len, out);
}
// header flags of a batch blob (bit 0: synthetic shards), or -1 if it is not a batch blob
int zktls_batch_flags(const uint8_t* blob, size_t len) {
    std::vector<std::vector<uint8_t>> proofs;
    uint32_t flags = 0;
    if (!zktls::unpack_shard_proofs(std::vector<uint8_t>(blob, blob + len), &proofs, &flags)) return -1;
    return (int)flags;
}
// splits a batch blob; returns the shard count or -1; offsets/lengths arrays sized `cap`
int zktls_unpack_batch(const uint8_t* blob, size_t len, size_t* offsets, size_t* lengths, int cap) {
    std::vector<std::vector<uint8_t>> proofs;
    if (!zktls::unpack_shard_proofs(std::vector<uint8_t>(blob, blob + len), &proofs)) return -1;
    size_t off = 16;
    for (size_t i = 0; i < proofs.size() && (int)i < cap; i++) { offsets[i] = off + 4; lengths[i] = proofs[i].size(); off += 4 + proofs[i].size(); }
    return (int)proofs.size();
}

}  // extern "C"
