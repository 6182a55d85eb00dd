// params.cpp -- the Poseidon2 parameter tables in effect, and their loader ("drop-in Poseidon2 tables
// ... so that dropping in real SP1 constants needs no rebuild").
//
// The reference's provers take their round constants from p3-poseidon2 / sp1-stark (BabyBearPoseidon2, reference
// Cargo.lock:4030, 6172) and risc0-zkp (Cargo.lock:5057); those tables are not obtainable offline, so the library ships its own
// generated set (p2_params.h) and reads any other set from a parameter file.  One set per process (it defines what a commitment
// means, for provers and verifier alike); it can only be changed while no context exists, and every context uploads the set in
// effect to its device when it is created.  File format: the JSON of tests/golden/poseidon2*_params.json -- keys "width"
// (16 or 24), "name", "external_rc" (8 x width), "internal_rc" (13 / 21), "internal_diag" (width); canonical residues.  The
// shape of the permutation (x^7, 8 full rounds, 13 / 21 partial rounds, the M4 blocks) is code, not data.
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "context.h"
#include "poseidon2.cuh"

namespace zk {

P2Tables g_p2_tables = P2_BUILTIN;
static P2Raw g_p2_raw = P2_BUILTIN_RAW;
static std::string g_name16 = "zktls-amd/p2-bb16-v1", g_name24 = "zktls-amd/p2-bb24-v1";
static std::mutex g_params_mu;
std::atomic<int> g_live_contexts{0};
std::atomic<uint64_t> g_p2_generation{0};     // bumped whenever the table set changes (caches of values derived from it check it)

// every integer of the (possibly nested) JSON array that follows "key":
static bool json_array(const std::string& text, const char* key, std::vector<uint64_t>& out) {
    const std::string k = std::string("\"") + key + "\"";
    size_t i = text.find(k);
    if (i == std::string::npos) return false;
    i = text.find('[', i);
    if (i == std::string::npos) return false;
    int depth = 0;
    out.clear();
    for (; i < text.size(); i++) {
        const char c = text[i];
        if (c == '[') depth++;
        else if (c == ']') { if (--depth == 0) return true; }
        else if (std::isdigit((unsigned char)c)) {
            uint64_t v = 0;
            while (i < text.size() && std::isdigit((unsigned char)text[i])) { v = v * 10 + (uint64_t)(text[i] - '0'); if (v > 0xFFFFFFFFull) return false; i++; }
            i--;
            out.push_back(v);
        } else if (c == '-' || c == '.' || c == 'e' || c == 'E') return false;     // residues are plain non-negative integers
    }
    return false;
}
static bool json_int(const std::string& text, const char* key, uint64_t& out) {
    const std::string k = std::string("\"") + key + "\"";
    size_t i = text.find(k);
    if (i == std::string::npos) return false;
    i = text.find(':', i);
    if (i == std::string::npos) return false;
    i++;
    while (i < text.size() && std::isspace((unsigned char)text[i])) i++;
    if (i >= text.size() || !std::isdigit((unsigned char)text[i])) return false;
    out = 0;
    while (i < text.size() && std::isdigit((unsigned char)text[i])) { out = out * 10 + (uint64_t)(text[i] - '0'); i++; }
    return true;
}
static std::string json_string(const std::string& text, const char* key) {
    const std::string k = std::string("\"") + key + "\"";
    size_t i = text.find(k);
    if (i == std::string::npos) return "";
    i = text.find(':', i);
    if (i == std::string::npos) return "";
    i = text.find('"', i);
    if (i == std::string::npos) return "";
    const size_t j = text.find('"', i + 1);
    return j == std::string::npos ? "" : text.substr(i + 1, j - i - 1);
}

}  // namespace zk

using namespace zk;

extern "C" {

// how often the table set changed in this process: holders of values derived from the tables (a parked machine key ...) compare it
uint64_t zkhip_poseidon2_params_generation(void) { return zk::g_p2_generation.load(); }
int zkhip_load_poseidon2_params(const char* path) {
    if (!path) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: null path");
    std::lock_guard<std::mutex> lk(g_params_mu);
    if (g_live_contexts.load() != 0)
        return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: contexts exist (destroy them and call zkhip_release_cached_contexts first): "
                                       "the parameter set is uploaded to a device when a context is created");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(ZKHIP_ERR_INVALID, std::string("load_poseidon2_params: cannot open ") + path);
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    uint64_t width = 0, p = 0, rf = 0, rp = 0, deg = 0;
    if (!json_int(text, "width", width) || (width != 16 && width != 24)) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: \"width\" must be 16 or 24");
    if (json_int(text, "p", p) && p != P) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: \"p\" is not the BabyBear prime");
    if (json_int(text, "rounds_f", rf) && rf != 8) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: 8 full rounds are compiled in");
    if (json_int(text, "rounds_p", rp) && rp != (width == 16 ? 13u : 21u)) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: 13 (width 16) / 21 (width 24) partial rounds are compiled in");
    if (json_int(text, "sbox_degree", deg) && deg != 7) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: the S-box x^7 is compiled in");
    std::vector<uint64_t> ext, inr, diag;
    if (!json_array(text, "external_rc", ext) || !json_array(text, "internal_rc", inr) || !json_array(text, "internal_diag", diag))
        return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: missing or malformed external_rc / internal_rc / internal_diag");
    const size_t n_int = width == 16 ? 13 : 21;
    if (ext.size() != 8 * width || inr.size() != n_int || diag.size() != width)
        return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: table sizes must be 8 x width, 13 / 21, width");
    for (uint64_t v : ext) if (v >= P) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: non-canonical residue");
    for (uint64_t v : inr) if (v >= P) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: non-canonical residue");
    for (uint64_t v : diag) if (v >= P) return fail(ZKHIP_ERR_INVALID, "load_poseidon2_params: non-canonical residue");
    P2Raw raw = g_p2_raw;
    if (width == 16) {
        for (int r = 0; r < 8; r++) for (int i = 0; i < 16; i++) raw.ext16[r][i] = (uint32_t)ext[16 * r + i];
        for (int r = 0; r < 13; r++) raw.int16[r] = (uint32_t)inr[r];
        for (int i = 0; i < 16; i++) raw.diag16[i] = (uint32_t)diag[i];
    } else {
        for (int r = 0; r < 8; r++) for (int i = 0; i < 24; i++) raw.ext24[r][i] = (uint32_t)ext[24 * r + i];
        for (int r = 0; r < 21; r++) raw.int24[r] = (uint32_t)inr[r];
        for (int i = 0; i < 24; i++) raw.diag24[i] = (uint32_t)diag[i];
    }
    const P2Tables t = derive_tables(raw);
    if (!tables_ok(raw, t)) return fail(ZKHIP_ERR_INTERNAL, "load_poseidon2_params: derived tables failed their self-check");
    g_p2_raw = raw;
    g_p2_tables = t;
    g_p2_generation.fetch_add(1);
    const std::string name = json_string(text, "name");
    (width == 16 ? g_name16 : g_name24) = name.empty() ? std::string(path) : name;
    return ZKHIP_OK;
}

int zkhip_reset_poseidon2_params(void) {
    std::lock_guard<std::mutex> lk(g_params_mu);
    if (g_live_contexts.load() != 0) return fail(ZKHIP_ERR_INVALID, "reset_poseidon2_params: contexts exist");
    g_p2_raw = P2_BUILTIN_RAW;
    g_p2_tables = P2_BUILTIN;
    g_p2_generation.fetch_add(1);
    g_name16 = "zktls-amd/p2-bb16-v1"; g_name24 = "zktls-amd/p2-bb24-v1";
    return ZKHIP_OK;
}

const char* zkhip_poseidon2_params_name(int width) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_params_mu);
    copy = width == 24 ? g_name24 : g_name16;
    return copy.c_str();
}

}  // extern "C"
