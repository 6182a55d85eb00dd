// moongate_hip.cpp -- a prover SERVER in front of libzkhip, speaking the transport the reference's SP1 path already knows how to
// call (SURVEY.md section 8b plug point 4, 8f-3).
//
// When `--sp1-moongate-server` / SP1_MOONGATE_SERVER is set (bins/zktls/src/commands/prove.rs:45-47) the reference builds
// `ProverClient::builder().cuda().with_moongate_endpoint(server)` (crates/guest-prover-sp1/src/sp1.rs:86-90) and every GPU step
// happens in another process over HTTP: sp1-cuda 4.1.4 (reference Cargo.lock:5878) is a twirp client (twirp-rs, Cargo.lock:6835).
// A server on that endpoint is a drop-in with no Rust at all.  What is implemented here, and what is not:
//   * the TRANSPORT, as the public Twirp v7 specification defines it: HTTP/1.1 POST /twirp/<package>.<Service>/<Method>,
//     Content-Type application/protobuf, the request and response being one protobuf message; errors as JSON
//     {"code", "msg"} with Twirp's HTTP status mapping (bad_route 404, malformed 400, unimplemented 501, unavailable 503, internal 500);
//   * the service SHAPE [RECALLED from sp1-cuda; its .proto is not in /root/reference]: package `api`, service `ProverService`,
//     methods Ready {} -> {bool ready = 1}, Setup / ProveCore / Compress / Shrink / Wrap {bytes data = 1} -> {bytes result = 1};
//   * the PAYLOADS inside `data` upstream are bincode serialisations of sp1-prover types (proving key, stdin, shard proofs) that
//     cannot be reproduced offline, so this server defines its own for the two methods it can serve:
//       Setup      data = guest ELF bytes                   -> result = 32 bytes: zkhip_request_digest("", ELF) (a stand-in "vk digest")
//       ProveCore  data = "ZKMG" u32 version(1) | i32 log_n | u32 width | u32 shards | i32 num_queries | i32 pow_bits |
//                         u32 backend (0 SP1 shape, 1 RISC Zero shape) | i32 device (-1: all) | u32 cbor_len | cbor | u32 elf_len | elf
//                                                            -> result = u32 output length | output | the batch blob of the host mirror
//                                                               (zktls_amd/host): flagged SYNTHETIC for shards > 0; shards = 0 asks for the
//                                                               input-commitment guest (SHA-256 chip over the CBOR input, flag INPUT_SHA256,
//                                                               output = the digest; log_n / width are ignored)
//                  version 2 = version 1 + u32 flags after `device` (bit 0 KEYED: shards must be 0; the mirror's setup() runs first -- the
//                         reference's setup -> prove -> verify, sp1.rs:113-120 -- and the commitment guest is proven as the keyed SHA-256
//                         machine)       -> result = u32 output length | output | u32 vk length | vk (64 bytes, or 0) | batch blob
//                         bit 1 COMPRESS (version 2): core -> compress behind the one call (HipGuestProver::with_compress, sp1.rs:116): the blob
//                         carries ONE proof that verifies the shard proofs in-circuit -- flags COMPRESSED (| TREE when the joins were joined again)
//       Compress   data = "ZKMC" u32 version(1) | i32 log_n | u32 width | u32 shards | i32 num_queries | i32 pow_bits | i32 device (-1: the first) |
//                         u32 shard proofs per join (0: as many as one join holds; fewer: several joins and ONE proof above them, the tree) |
//                         u32 cbor_len | cbor | u32 elf_len | elf | u32 blob_len | the batch blob ProveCore returned for that plan, input and ELF
//                                                            -> result = the compressed batch blob (zktls::compress_blob): sp1-cuda's compress step
//                                                               as its own call; zktls_verify_compressed_blob checks it from (plan, input, ELF, key)
//       Shrink / Wrap                                        -> Twirp error `unimplemented` (the BN254 wrap + Groth16 stages are out of scope)
//     Swapping in upstream's payload structs is the remaining work once they can be read; the transport does not change.
// One request at a time per connection, connections served one after the other (proofs serialise on the GPU anyway); 127.0.0.1 only
// unless --bind is given.  No TLS, no auth: it is meant to sit next to the client like the container it replaces.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"
#include "../../include/zkhip_hal.h"
#include "../host/guest_prover_hip.hpp"

namespace {

struct Request { std::string method, path, content_type; std::vector<uint8_t> body; bool ok = false; };

bool write_all(int fd, const void* buf, size_t n) {
    const uint8_t* p = (const uint8_t*)buf;
    while (n) { ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL); if (r <= 0) return false; p += r; n -= (size_t)r; }
    return true;
}
std::string lower(std::string s) { for (char& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c + 32); return s; }

// minimal HTTP/1.1 request reader: request line, headers (Content-Length, Content-Type), body.  The caller has set receive / send
// timeouts on the socket, so a stalled peer costs the accept loop a bounded wait.  Headers arrive through a buffer (not a recv per
// byte); the body is read in chunks into a vector that grows with the bytes that actually arrive, so a large Content-Length
// alone allocates nothing.
constexpr size_t MAX_HEAD = 16384;
constexpr size_t BODY_CHUNK = (size_t)1 << 20;
Request read_request(int fd, size_t max_body) {
    Request rq;
    std::string head;
    std::vector<uint8_t> extra;                       // bytes received past the header terminator: the start of the body
    {
        char buf[4096];
        size_t end = std::string::npos;
        while (head.size() < MAX_HEAD) {
            const ssize_t r = ::recv(fd, buf, sizeof buf, 0);
            if (r <= 0) return rq;
            const size_t from = head.size() >= 3 ? head.size() - 3 : 0;
            head.append(buf, (size_t)r);
            end = head.find("\r\n\r\n", from);
            if (end != std::string::npos) break;
        }
        if (end == std::string::npos) return rq;
        extra.assign(head.begin() + (long)end + 4, head.end());
        head.resize(end + 4);
    }
    const size_t eol = head.find("\r\n");
    if (eol == std::string::npos) return rq;
    const std::string line = head.substr(0, eol);
    const size_t s1 = line.find(' '), s2 = line.rfind(' ');
    if (s1 == std::string::npos || s2 == s1) return rq;
    rq.method = line.substr(0, s1);
    rq.path = line.substr(s1 + 1, s2 - s1 - 1);
    size_t content_length = 0;
    size_t pos = eol + 2;
    while (pos < head.size()) {
        const size_t e = head.find("\r\n", pos);
        if (e == std::string::npos || e == pos) break;
        const std::string h = head.substr(pos, e - pos);
        const size_t colon = h.find(':');
        if (colon != std::string::npos) {
            const std::string k = lower(h.substr(0, colon));
            std::string v = h.substr(colon + 1);
            while (!v.empty() && v[0] == ' ') v.erase(0, 1);
            if (k == "content-length") content_length = (size_t)std::strtoull(v.c_str(), nullptr, 10);
            if (k == "content-type") rq.content_type = lower(v);
        }
        pos = e + 2;
    }
    if (content_length > max_body || extra.size() > content_length) return rq;
    rq.body = std::move(extra);
    while (rq.body.size() < content_length) {
        const size_t have = rq.body.size();
        const size_t want = content_length - have < BODY_CHUNK ? content_length - have : BODY_CHUNK;
        rq.body.resize(have + want);
        const ssize_t r = ::recv(fd, rq.body.data() + have, want, 0);
        if (r <= 0) { rq.body.clear(); return rq; }
        rq.body.resize(have + (size_t)r);
    }
    rq.ok = true;
    return rq;
}

void respond(int fd, int status, const char* reason, const char* ctype, const void* body, size_t n) {
    char head[256];
    const int hl = std::snprintf(head, sizeof head, "HTTP/1.1 %d %s\r\nContent-Type: %s\r\nContent-Length: %zu\r\nConnection: close\r\n\r\n", status, reason, ctype, n);
    if (write_all(fd, head, (size_t)hl) && n) write_all(fd, body, n);
}
std::string json_escape(const std::string& s) {
    std::string o;
    for (char c : s) { if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); } else if ((unsigned char)c < 0x20) o.push_back(' '); else o.push_back(c); }
    return o;
}
// Twirp error: JSON body, status by error code
void twirp_error(int fd, const char* code, const std::string& msg) {
    int status = 500; const char* reason = "Internal Server Error";
    if (!std::strcmp(code, "bad_route")) { status = 404; reason = "Not Found"; }
    else if (!std::strcmp(code, "malformed") || !std::strcmp(code, "invalid_argument")) { status = 400; reason = "Bad Request"; }
    else if (!std::strcmp(code, "unimplemented")) { status = 501; reason = "Not Implemented"; }
    else if (!std::strcmp(code, "unavailable")) { status = 503; reason = "Service Unavailable"; }
    const std::string body = std::string("{\"code\":\"") + code + "\",\"msg\":\"" + json_escape(msg) + "\"}";
    respond(fd, status, reason, "application/json", body.data(), body.size());
}

// protobuf: the only messages are {bytes f1 = 1} and {bool f1 = 1}
bool pb_get_bytes1(const std::vector<uint8_t>& m, std::vector<uint8_t>* out) {
    size_t p = 0;
    out->clear();
    while (p < m.size()) {
        uint64_t key = 0; int sh = 0;
        for (;;) { if (p >= m.size() || sh > 63) return false; const uint8_t b = m[p++]; key |= (uint64_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; }
        const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (wt == 0) { for (;;) { if (p >= m.size()) return false; if (!(m[p++] & 0x80)) break; } }
        else if (wt == 2) {
            uint64_t len = 0; sh = 0;
            for (;;) { if (p >= m.size() || sh > 63) return false; const uint8_t b = m[p++]; len |= (uint64_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; }
            if (len > m.size() - p) return false;
            if (field == 1) out->assign(m.begin() + (long)p, m.begin() + (long)(p + len));
            p += len;
        } else if (wt == 5) { if (m.size() - p < 4) return false; p += 4; }
        else if (wt == 1) { if (m.size() - p < 8) return false; p += 8; }
        else return false;
    }
    return true;
}
std::vector<uint8_t> pb_bytes1(const std::vector<uint8_t>& v) {
    std::vector<uint8_t> o;
    o.push_back(0x0A);
    uint64_t n = v.size();
    do { uint8_t b = n & 0x7F; n >>= 7; if (n) b |= 0x80; o.push_back(b); } while (n);
    o.insert(o.end(), v.begin(), v.end());
    return o;
}

struct Cursor {
    const std::vector<uint8_t>& d; size_t p = 0; bool ok = true;
    uint32_t u32() { if (d.size() - p < 4) { ok = false; return 0; } uint32_t v; std::memcpy(&v, d.data() + p, 4); p += 4; return v; }
    std::vector<uint8_t> blob() { const uint32_t n = u32(); if (!ok || d.size() - p < n) { ok = false; return {}; } std::vector<uint8_t> v(d.begin() + (long)p, d.begin() + (long)(p + n)); p += n; return v; }
};

void handle(int fd) {
    // a request carries one CBOR input and one ELF (tens of KB to a few MB) and, for Compress, a batch of shard proofs (64 of the headline
    // shape are 61 MB): 256 MiB is generous, and the body buffer grows only with bytes that arrive
    const Request rq = read_request(fd, (size_t)256 << 20);
    if (!rq.ok) { twirp_error(fd, "malformed", "could not read an HTTP request"); return; }
    const std::string prefix = "/twirp/api.ProverService/";
    if (rq.method != "POST") { twirp_error(fd, "bad_route", "unsupported method " + rq.method + " (only POST is allowed)"); return; }
    if (rq.path.compare(0, prefix.size(), prefix) != 0) { twirp_error(fd, "bad_route", "no handler for path " + rq.path); return; }
    if (rq.content_type.compare(0, 20, "application/protobuf") != 0) { twirp_error(fd, "bad_route", "unexpected Content-Type: " + rq.content_type + " (this server speaks application/protobuf)"); return; }
    const std::string method = rq.path.substr(prefix.size());
    if (method == "Ready") {
        const uint8_t yes[2] = {0x08, 0x01};
        if (zkhip_device_count() > 0) respond(fd, 200, "OK", "application/protobuf", yes, 2);
        else respond(fd, 200, "OK", "application/protobuf", nullptr, 0);        // proto3: false is the empty message
        return;
    }
    if (method == "Shrink" || method == "Wrap") { twirp_error(fd, "unimplemented", method + ": the BN254 wrap and Groth16 stages are not part of the shard-prove hot path"); return; }
    if (method != "Setup" && method != "ProveCore" && method != "Compress") { twirp_error(fd, "bad_route", "no handler for path " + rq.path); return; }
    std::vector<uint8_t> data;
    if (!pb_get_bytes1(rq.body, &data)) { twirp_error(fd, "malformed", "the request is not a protobuf message with a bytes field 1"); return; }
    if (method == "Setup") {
        if (data.empty()) { twirp_error(fd, "invalid_argument", "Setup: empty program"); return; }
        uint32_t dg[8];
        if (zkhip_request_digest(nullptr, 0, data.data(), data.size(), dg) != ZKHIP_OK) { twirp_error(fd, "internal", zkhip_last_error()); return; }
        std::vector<uint8_t> res(32);
        std::memcpy(res.data(), dg, 32);
        const std::vector<uint8_t> msg = pb_bytes1(res);
        respond(fd, 200, "OK", "application/protobuf", msg.data(), msg.size());
        return;
    }
    // the plan comes from an unauthenticated peer: bound every field before anything is sized by it (advice r2: shards = 2^32 - 1
    // value-initialised ~100 GB of proof slots).  Ranges are the library's own (include/zkhip.h) with a batch cap on top.
    auto bad_plan = [](const zktls::ShardPlan& plan, int device) -> std::string {
        const int ndev = zkhip_device_count();
        if (plan.shards > 4096u) return "shards must be in [0, 4096]";
        if (plan.shards != 0 && (plan.log_n < 5 || plan.log_n > 22)) return "log_n must be in [5, 22]";
        if (plan.shards != 0 && (plan.width < 4u || plan.width > 1024u)) return "width must be in [4, 1024]";
        if (plan.num_queries < 1 || plan.num_queries > 256) return "num_queries must be in [1, 256]";
        if (plan.pow_bits < 0 || plan.pow_bits > 24) return "pow_bits must be in [0, 24]";
        if (device < -1 || device >= (ndev > 0 ? ndev : 1)) return "device must be -1 (every GPU) or a visible device ordinal";
        if (plan.shards != 0 && ((uint64_t)plan.shards << plan.log_n) * plan.width > ((uint64_t)1 << 36)) return "the batch exceeds 2^36 trace cells";
        return "";
    };
    auto unavailable = [](const std::string& e) { return e.find("no CPU fallback") != std::string::npos || e.find("NO_DEVICE") != std::string::npos || e.find("no device") != std::string::npos; };
    Cursor c{data};
    if (method == "Compress") {
        const uint32_t magic = c.u32(), version = c.u32();
        if (magic != 0x434D4B5Au || version != 1u) { twirp_error(fd, "invalid_argument", "Compress: payload must start with \"ZKMC\", version 1 (see moongate_hip.cpp)"); return; }
        zktls::ShardPlan plan;
        plan.log_n = (int)c.u32(); plan.width = c.u32(); plan.shards = c.u32(); plan.num_queries = (int)c.u32(); plan.pow_bits = (int)c.u32();
        const int device = (int)c.u32();
        const uint32_t per_join = c.u32();
        const std::vector<uint8_t> cbor = c.blob(), elf = c.blob(), blob = c.blob();
        if (!c.ok || c.p != data.size()) { twirp_error(fd, "invalid_argument", "Compress: truncated or oversized payload"); return; }
        std::string bad = bad_plan(plan, device);
        if (bad.empty() && plan.shards == 0) bad = "shards must be at least 1";
        if (bad.empty() && per_join > 4096u) bad = "shard proofs per join must be in [0, 4096]";
        if (!bad.empty()) { twirp_error(fd, "invalid_argument", "Compress: " + bad); return; }
        if (zkhip_device_count() <= 0) { twirp_error(fd, "unavailable", "Compress: no device (there is no CPU fallback)"); return; }
        zktls::set_compress_join_size(per_join);                                           // (process-wide, one request at a time: part of the plan)
        const zktls::ProveResult r = zktls::compress_blob(device < 0 ? 0 : device, plan, cbor, elf, blob);
        zktls::set_compress_join_size(0);
        if (!r.ok) { twirp_error(fd, unavailable(r.error) ? "unavailable" : "invalid_argument", r.error); return; }
        const std::vector<uint8_t> msg = pb_bytes1(r.proof);
        respond(fd, 200, "OK", "application/protobuf", msg.data(), msg.size());
        return;
    }
    const uint32_t magic = c.u32(), version = c.u32();
    if (magic != 0x474D4B5Au || (version != 1u && version != 2u)) { twirp_error(fd, "invalid_argument", "ProveCore: payload must start with \"ZKMG\", version 1 or 2 (see moongate_hip.cpp)"); return; }
    zktls::ShardPlan plan;
    plan.log_n = (int)c.u32(); plan.width = c.u32(); plan.shards = c.u32(); plan.num_queries = (int)c.u32(); plan.pow_bits = (int)c.u32();
    const uint32_t backend = c.u32();
    const int device = (int)c.u32();
    const uint32_t flags = version >= 2 ? c.u32() : 0u;
    zktls::GuestInput in;
    in.cbor = c.blob();
    const std::vector<uint8_t> elf = c.blob();
    if (!c.ok || c.p != data.size() || backend > 1 || flags > 3) { twirp_error(fd, "invalid_argument", "ProveCore: truncated or oversized payload"); return; }
    {
        const std::string bad = bad_plan(plan, device);
        if (!bad.empty()) { twirp_error(fd, "invalid_argument", "ProveCore: " + bad); return; }
    }
    if ((flags & 1u) && (plan.shards != 0 || backend != 0)) { twirp_error(fd, "invalid_argument", "ProveCore: KEYED asks for the input-commitment guest (shards = 0) in the SP1 shape"); return; }
    zktls::HipGuestProver prover(device < 0 ? 0 : device, backend ? zktls::Backend::Risc0 : zktls::Backend::Sp1);
    if (device < 0) { std::vector<int> all; for (int d = 0; d < zkhip_device_count(); d++) all.push_back(d); if (!all.empty()) prover.with_devices(all); }
    if ((flags & 2u) && backend != 0) { twirp_error(fd, "invalid_argument", "ProveCore: COMPRESS takes SP1-shape shard proofs"); return; }
    if ((flags & 3u) == 3u) { twirp_error(fd, "invalid_argument", "ProveCore: KEYED and COMPRESS do not combine"); return; }
    if (plan.shards == 0) prover.hip().with_input_commitment(plan);
    else prover.hip().with_synthetic(plan);
    if (flags & 2u) prover.with_compress();
    if (flags & 1u) {
        const zktls::SetupResult s = prover.setup(elf);                                // sp1.rs:113
        if (!s.ok) {
            const bool nodev = s.error.find("no CPU fallback") != std::string::npos || s.error.find("NO_DEVICE") != std::string::npos;
            twirp_error(fd, nodev ? "unavailable" : "invalid_argument", s.error);
            return;
        }
    }
    const zktls::ProveResult r = prover.prove(in, elf);
    if (!r.ok) {
        const bool nodev = r.error.find("no CPU fallback") != std::string::npos || r.error.find("NO_DEVICE") != std::string::npos;
        twirp_error(fd, nodev ? "unavailable" : "invalid_argument", r.error);
        return;
    }
    // result = u32 output length | output | proof blob
    std::vector<uint8_t> res;
    const uint32_t on = (uint32_t)r.output.size();
    res.insert(res.end(), (const uint8_t*)&on, (const uint8_t*)&on + 4);
    res.insert(res.end(), r.output.begin(), r.output.end());
    if (version >= 2) {
        const uint32_t vn = (uint32_t)r.vk.size();
        res.insert(res.end(), (const uint8_t*)&vn, (const uint8_t*)&vn + 4);
        res.insert(res.end(), r.vk.begin(), r.vk.end());
    }
    res.insert(res.end(), r.proof.begin(), r.proof.end());
    const std::vector<uint8_t> msg = pb_bytes1(res);
    respond(fd, 200, "OK", "application/protobuf", msg.data(), msg.size());
}

}  // namespace

int main(int argc, char** argv) {
    int port = 3000;
    const char* bind_addr = "127.0.0.1";
    int max_requests = -1;                                   // tests: serve this many requests, then exit
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--port") && i + 1 < argc) port = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--bind") && i + 1 < argc) bind_addr = argv[++i];
        else if (!std::strcmp(argv[i], "--max-requests") && i + 1 < argc) max_requests = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "usage: %s [--port N] [--bind ADDR] [--max-requests N]\n", argv[0]); return 2; }
    }
    signal(SIGPIPE, SIG_IGN);
    const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) { std::perror("socket"); return 1; }
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, bind_addr, &addr.sin_addr) != 1) { std::fprintf(stderr, "bad bind address\n"); return 2; }
    if (::bind(ls, (sockaddr*)&addr, sizeof addr) != 0 || ::listen(ls, 16) != 0) { std::perror("bind/listen"); return 1; }
    std::fprintf(stderr, "moongate-hip: listening on %s:%d (devices: %d)\n", bind_addr, port, zkhip_device_count());
    std::fflush(stderr);
    for (int served = 0; max_requests < 0 || served < max_requests; served++) {
        const int fd = ::accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        {   // one stalled connection must not hold the (single) accept loop
            timeval tv{};
            tv.tv_sec = 20;
            setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
        }
        handle(fd);
        ::shutdown(fd, SHUT_RDWR);
        ::close(fd);
    }
    zktls::release_cached();
    zkhip_release_cached_contexts();
    ::close(ls);
    return 0;
}
