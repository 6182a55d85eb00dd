// sha256_chip.hip -- a real chip AIR on the constraint-program path: SHA-256 compression, the hash of the
// TLS transcripts the reference's guest program checks (the guest ELF itself is not in /root/reference; upstream, SP1 proves
// SHA-256 through its ShaExtend / ShaCompress precompile chips: sp1-core-machine 4.1.4, reference Cargo.lock:5822, behind
// crates/guest-prover-sp1/src/sp1.rs:116).  This is not SP1's chip pair (those exchange memory through lookups with the CPU chip);
// it is a self-contained AIR for the same function, written out as a constraint program (air.h) plus its on-device trace
// generator, proven and verified by zkhip_prove_shard_air / zkhip_verify_shard_air like any other program.
//
// One row per round, 64 rows per 64-byte block, blocks one after the other; 640 columns (612 in use), every constraint of degree <= 3.
// A proof says: "I know at most 2^k blocks whose SHA-256 chaining value, from the standard IV, is the 16 public 16-bit limbs"
// -- with FIPS 180-4 padding inside the blocks, the SHA-256 digest of a message.  Blocks after the message are INACTIVE and
// pass the chaining value through (the trace height is a power of two, a block count is not).
//
// Columns (bit i of a word = base + i, least significant first; a limb pair = low 16 bits, high 16 bits):
//   SEL 64 one-hot round selector | A B C E F G 32 bits each (working variables before the round) | D HV limb pairs (d, h)
//   S1 CH S0 MJ 32 bits each (Sigma1(e), Ch, Sigma0(a), Maj) | HC 8 limb pairs (chaining value of the block)
//   OUT 8 limb pairs (variables after the round, + chaining value in round 63, mod 2^32) | X0 X13 32 bits (W_t, W_{t+13})
//   XL 14 limb pairs (W_{t+j}, j = 1..12, 14, 15) | SG0 SG1 32 bits (sigma0(W_t), sigma1(W_{t+13})) | CY 28 carry bits
//   ACT (block belongs to the message) | SKIP = s_63 (1 - ACT)
//   CNT (active blocks left, this one included) | LASTB (the last active block) | L2 (the block before it) | SB (the row whose W_t is the
//   word that holds the 0x80 byte) | Z0 = s_0 LASTB | Z2 = s_0 L2                                   -- the padding columns, round 5
//
// PADDING IN-CIRCUIT (round 5).  Until round 4 a proof said "a compression chain over SOME block sequence ends in this digest": ACT could
// drop after any block, the 0x80 byte and the length field were the host's business.  Now the statement is "digest = SHA-256 of a message of
// L bytes" with L PUBLIC: the verifier derives 75 more public values from L (sha::padding_publics: the block count K, where the 0x80 byte
// sits -- block, word, byte --, which words must be zero, the length field) and the program pins
//   * the block count: CNT = K on the first row, CNT' = CNT - s_63 ACT, (1 - ACT) CNT = 0, CNT = ACT on the last row -- ACT can neither drop
//     early (CNT would not be 0 yet) nor late (CNT would pass 0);
//   * LASTB / L2: block-constant flags; ACT - ACT' = LASTB where a block ends, LASTB = ACT on the last row, LASTB' = L2 where a block ends;
//   * the boundary word: SB = sum_j (BWL_j LASTB + BW2_j L2) s_j selects the ROW t = j of the block with the 0x80 byte, where X0 holds the
//     word's 32 bits: below the byte's position the bits are fixed (1 then zeros: KIND_c picks one of the four byte positions), above it
//     they are the message's;
//   * the words that must be zero and the 64-bit length field, at row s_0 of their block, where all sixteen words of the block are in the
//     window (X0, X13 as bits, the others as limb pairs): Z0 / Z2 times a public mask times the limb.
// Every term keeps to three factors (public values count: air.h), so the quotient degree stays 2 chunks.  A chained shard (a slice of a
// longer message) gets the same constraints with the public values of ITS slice: no boundary, no length field in a slice that has neither.
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <cstring>
#include <string>
#include <vector>

#include "air.h"
#include "context.h"
#include "batch.h"
namespace zk { int lockstep_selftest_observe(int members); }      // prover.cpp: the members' transcripts absorbing side by side (LaunchBatcher::host_merge) against one after the other

namespace zk {
namespace sha {

constexpr uint32_t SEL = 0, A = 64, B = 96, C = 128, E = 160, F = 192, G = 224, D = 256, HV = 258;
constexpr uint32_t S1 = 260, CH = 292, S0 = 324, MJ = 356, HC = 388, OUT = 404, X0 = 420, X13 = 452, XL = 484;
constexpr uint32_t SG0 = 512, SG1 = 544, CY = 576, ACT = 604, SKIP = 605, CNT = 606, LASTB = 607, L2 = 608, SB = 609, Z0 = 610, Z2 = 611, USED = 612, WIDTH = 640;
// (612 columns in use; 640 = 20 tiles of 32 columns: the LDE's two-columns-per-lane passes and its fused middle launch take whole tiles only --
// a 612-wide matrix costs 9.7 ms per 2^20-row LDE against 6.3 at 640, and the leaf hash pays 80 permutations per row instead of 77)
// public values: the digest's 16 limbs (chained: then the initial chaining value's 16), then the padding's:
constexpr uint32_t N_DIGEST = 16, PP_K = 0, PP_FIN = 1, PP_Z13 = 2, PP_BWL = 3, PP_BW2 = 19, PP_KIND = 35, PP_ZWL = 39, PP_ZW2 = 55, PP_LEN = 71, N_PAD = 75;
constexpr uint32_t N_PUBLIC = N_DIGEST + N_PAD, N_PUBLIC_CHAINED = 2 * N_DIGEST + N_PAD;
constexpr uint32_t CY_A = CY, CY_E = CY + 6, CY_W6 = CY + 12, CY_SCHED = CY + 24;

constexpr uint32_t IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

// K_t = first 32 bits of the fractional part of the cube root of the t-th prime (FIPS 180-4, 4.2.2), from integers
struct RoundConstants {
    uint32_t k[64];
    RoundConstants() {
        int found = 0;
        for (uint32_t n = 2; found < 64; n++) {
            bool prime = true;
            for (uint32_t q = 2; q * q <= n; q++) if (n % q == 0) { prime = false; break; }
            if (!prime) continue;
            unsigned __int128 lo = 0, hi = (unsigned __int128)1 << 40;           // floor(cbrt(n) 2^32) by bisection
            while (hi - lo > 1) {
                const unsigned __int128 mid = (lo + hi) / 2;
                if (mid * mid * mid <= ((unsigned __int128)n << 96)) lo = mid; else hi = mid;
            }
            k[found++] = (uint32_t)lo;
        }
    }
};
static const RoundConstants& round_constants() { static const RoundConstants rc; return rc; }

__host__ __device__ inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
__host__ __device__ inline uint32_t big_sigma0(uint32_t a) { return rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22); }
__host__ __device__ inline uint32_t big_sigma1(uint32_t e) { return rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25); }
__host__ __device__ inline uint32_t small_sigma0(uint32_t w) { return rotr(w, 7) ^ rotr(w, 18) ^ (w >> 3); }
__host__ __device__ inline uint32_t small_sigma1(uint32_t w) { return rotr(w, 17) ^ rotr(w, 19) ^ (w >> 10); }

// ---- what the padding of a message of L bytes means for a trace that holds blocks [first, first + n_active) of the padded message ----
// FIPS 180-4 5.1.1: the message, 0x80, zeros, the bit length as 64 bits big-endian; k = (L + 8) / 64 + 1 blocks; the 0x80 byte is byte
// L mod 64 of block q = L / 64 (q = k - 1, or k - 2 when L mod 64 >= 56).  out[N_PAD]: the public values the program's padding constraints
// read (layout PP_*), a function of (L, first, n_active) alone -- the verifier computes them, the proof does not carry them.
struct PadPlace { uint32_t pad_block = 0xFFFFFFFFu, pad_row = 0; };       // trace-relative block with the 0x80 byte and the row (= word index) to flag
inline PadPlace padding_publics(uint64_t L, uint64_t first, uint64_t n_active, uint32_t out[N_PAD]) {
    for (uint32_t i = 0; i < N_PAD; i++) out[i] = 0u;
    PadPlace pl;
    const uint64_t k = (L + 8) / 64 + 1, q = L / 64, r = L % 64;
    const bool has_last = n_active > 0 && k - 1 >= first && k - 1 < first + n_active;
    const bool has_pad = n_active > 0 && q >= first && q < first + n_active;
    out[PP_K] = (uint32_t)n_active;
    out[PP_FIN] = has_last ? 1u : 0u;
    out[PP_Z13] = (has_last && q != k - 1) ? 1u : 0u;           // the length block holds nothing else: words 0 .. 13 are zero
    if (has_pad) {
        const uint32_t j = (uint32_t)(r / 4), c = (uint32_t)(r % 4);
        // the block with the 0x80 byte is this trace's LAST active block (flag LASTB), or the one before it (flag L2) when the length block follows it here
        const bool by_l2 = q != k - 1 && has_last;
        out[(by_l2 ? PP_BW2 : PP_BWL) + j] = 1u;
        out[PP_KIND + c] = 1u;
        const uint32_t upto = (q == k - 1) ? 13u : 15u;          // zeros run to the length field (same block) or to the block's end
        for (uint32_t w = j + 1; w <= upto; w++) out[(by_l2 ? PP_ZW2 : PP_ZWL) + w] = 1u;
        pl.pad_block = (uint32_t)(q - first); pl.pad_row = j;
    }
    if (has_last) {
        const uint64_t bits = L * 8;
        out[PP_LEN + 0] = (uint32_t)(bits & 0xffffu); out[PP_LEN + 1] = (uint32_t)((bits >> 16) & 0xffffu);       // W_15: low, high limb
        out[PP_LEN + 2] = (uint32_t)((bits >> 32) & 0xffffu); out[PP_LEN + 3] = (uint32_t)((bits >> 48) & 0xffffu);  // W_14
    }
    return pl;
}

// ---- the constraint program ------------------------------------------------------------------------------------------
struct Term { uint32_t coeff; std::vector<uint32_t> vars; };
typedef std::vector<Term> Terms;
inline uint32_t var(uint32_t col, bool next = false) { return next ? ((1u << 30) | col) : col; }
inline uint32_t pub(uint32_t idx) { return (2u << 30) | idx; }
inline uint32_t neg(uint32_t c) { return c ? P - c : 0u; }
__host__ __device__ inline uint32_t xl(int j) { return XL + 2 * (j >= 14 ? j - 2 : j - 1); }          // limb pair of W_{t+j}, j in 1..12, 14, 15

struct Word { bool bits; uint32_t base; };
inline Terms limb(Word w, int l, bool next = false) {
    Terms t;
    if (w.bits) for (int i = 0; i < 16; i++) t.push_back(Term{1u << i, {var(w.base + 16 * l + i, next)}});
    else t.push_back(Term{1u, {var(w.base + l, next)}});
    return t;
}
inline Terms negated(const Terms& in) { Terms t = in; for (Term& x : t) x.coeff = neg(x.coeff); return t; }
inline Terms times(const Terms& in, uint32_t v, bool negate) {
    Terms t = in;
    for (Term& x : t) { x.vars.insert(x.vars.begin(), v); if (negate) x.coeff = neg(x.coeff); }
    return t;
}
inline void append(Terms& a, const Terms& b) { a.insert(a.end(), b.begin(), b.end()); }
inline Terms xor3(uint32_t x, uint32_t y, uint32_t z) {
    return Terms{{1, {x}}, {1, {y}}, {1, {z}}, {P - 2, {x, y}}, {P - 2, {y, z}}, {P - 2, {x, z}}, {4, {x, y, z}}};
}
inline Terms xor2(uint32_t x, uint32_t y) { return Terms{{1, {x}}, {1, {y}}, {P - 2, {x, y}}}; }

struct Builder {
    std::vector<uint32_t> body;
    uint32_t count = 0;
    void add(uint32_t selector, const Terms& terms) {
        body.push_back(selector);
        body.push_back((uint32_t)terms.size());
        for (const Term& t : terms) {
            body.push_back(t.coeff % P);
            body.push_back((uint32_t)t.vars.size());
            for (uint32_t v : t.vars) body.push_back(v);
        }
        count++;
    }
};
enum : uint32_t { ALL = 0, FIRST = 1, LAST = 2, TRANSITION = 3 };

// chained = false: the chaining value of the first row is the standard IV (16 public values: the final chaining value's limbs);
// chained = true: it is PUBLIC too (32 public values: final limbs, then initial limbs) -- a shard of a longer message
static std::vector<uint32_t> build_program(bool chained) {
    {
        const uint32_t* K = round_constants().k;
        const Word words[8] = {{true, A}, {true, B}, {true, C}, {false, D}, {true, E}, {true, F}, {true, G}, {false, HV}};   // a .. h
        auto xw = [](int j) { return j == 0 ? Word{true, X0} : (j == 13 ? Word{true, X13} : Word{false, xl(j)}); };
        const uint32_t s63 = var(SEL + 63);
        Builder b;
        // round selector: s_0 = 1 on the first row, cyclic shift on transitions
        b.add(FIRST, Terms{{1, {var(SEL)}}, {P - 1, {}}});
        for (uint32_t t = 1; t < 64; t++) b.add(FIRST, Terms{{1, {var(SEL + t)}}});
        for (uint32_t t = 0; t < 64; t++) b.add(TRANSITION, Terms{{1, {var(SEL + (t + 1) % 64, true)}}, {P - 1, {var(SEL + t)}}});
        // bits are bits
        for (uint32_t base : {A, B, C, E, F, G, X0, X13})
            for (uint32_t i = 0; i < 32; i++) b.add(ALL, Terms{{1, {var(base + i), var(base + i)}}, {P - 1, {var(base + i)}}});
        for (uint32_t i = 0; i < 28; i++) b.add(ALL, Terms{{1, {var(CY + i), var(CY + i)}}, {P - 1, {var(CY + i)}}});
        // the bitwise functions of the round
        for (uint32_t i = 0; i < 32; i++) {
            Terms t{{1, {var(S1 + i)}}};
            append(t, negated(xor3(var(E + (i + 6) % 32), var(E + (i + 11) % 32), var(E + (i + 25) % 32))));
            b.add(ALL, t);
        }
        for (uint32_t i = 0; i < 32; i++)
            b.add(ALL, Terms{{1, {var(CH + i)}}, {P - 1, {var(G + i)}}, {P - 1, {var(E + i), var(F + i)}}, {1, {var(E + i), var(G + i)}}});
        for (uint32_t i = 0; i < 32; i++) {
            Terms t{{1, {var(S0 + i)}}};
            append(t, negated(xor3(var(A + (i + 2) % 32), var(A + (i + 13) % 32), var(A + (i + 22) % 32))));
            b.add(ALL, t);
        }
        for (uint32_t i = 0; i < 32; i++) {
            const uint32_t x = var(A + i), y = var(B + i), z = var(C + i);
            b.add(ALL, Terms{{1, {var(MJ + i)}}, {P - 1, {x, y}}, {P - 1, {x, z}}, {P - 1, {y, z}}, {2, {x, y, z}}});
        }
        // the schedule's bitwise functions: sigma0 = rotr 7 ^ rotr 18 ^ shr 3 of W_t, sigma1 = rotr 17 ^ rotr 19 ^ shr 10 of W_{t+13}
        const uint32_t sg[2][5] = {{SG0, X0, 7, 18, 3}, {SG1, X13, 17, 19, 10}};
        for (const auto& s : sg)
            for (uint32_t i = 0; i < 32; i++) {
                const uint32_t x = var(s[1] + (i + s[2]) % 32), y = var(s[1] + (i + s[3]) % 32);
                Terms t{{1, {var(s[0] + i)}}};
                append(t, negated(i + s[4] < 32 ? xor3(x, y, var(s[1] + i + s[4])) : xor2(x, y)));
                b.add(ALL, t);
            }
        // the round: OUT = (1 - SKIP) (new working variable) + s_63 (chaining value), mod 2^32, limb by limb with carry bits
        for (int l = 0; l < 2; l++) {
            Terms t1 = limb(Word{false, HV}, l);
            append(t1, limb(Word{true, S1}, l));
            append(t1, limb(Word{true, CH}, l));
            for (uint32_t t = 0; t < 64; t++) t1.push_back(Term{(K[t] >> (16 * l)) & 0xffffu, {var(SEL + t)}});
            append(t1, limb(Word{true, X0}, l));
            Terms new_a = t1;
            append(new_a, limb(Word{true, S0}, l));
            append(new_a, limb(Word{true, MJ}, l));
            Terms new_e = limb(Word{false, D}, l);
            append(new_e, t1);
            const Terms rhs[8] = {new_a, limb(words[0], l), limb(words[1], l), limb(words[2], l), new_e, limb(words[4], l), limb(words[5], l), limb(words[6], l)};
            const uint32_t cy[8] = {CY_A, CY_W6, CY_W6 + 2, CY_W6 + 4, CY_E, CY_W6 + 6, CY_W6 + 8, CY_W6 + 10};
            for (int w = 0; w < 8; w++) {
                const uint32_t ncy = (w == 0 || w == 4) ? 3 : 1;
                Terms t{{1, {var(OUT + 2 * w + l)}}};
                for (uint32_t k = 0; k < ncy; k++) t.push_back(Term{(uint32_t)(((uint64_t)1 << (16 + k)) % P), {var(cy[w] + ncy * l + k)}});
                append(t, negated(rhs[w]));
                append(t, times(rhs[w], var(SKIP), false));
                t.push_back(Term{P - 1, {s63, var(HC + 2 * w + l)}});
                if (l == 1) for (uint32_t k = 0; k < ncy; k++) t.push_back(Term{P - (1u << k), {var(cy[w] + k)}});     // carry out of the low limb
                b.add(ALL, t);
            }
        }
        // activity: a bit, 1 on the first row, never back from 0 to 1; SKIP = s_63 (1 - ACT)
        b.add(ALL, Terms{{1, {var(ACT), var(ACT)}}, {P - 1, {var(ACT)}}});
        b.add(FIRST, Terms{{1, {var(ACT)}}, {P - 1, {}}});
        b.add(TRANSITION, Terms{{1, {var(ACT, true)}}, {P - 1, {var(ACT, true), var(ACT)}}});
        b.add(ALL, Terms{{1, {var(SKIP)}}, {P - 1, {s63}}, {1, {s63, var(ACT)}}});
        // the next row starts from OUT
        for (int w = 0; w < 8; w++)
            for (int l = 0; l < 2; l++) {
                Terms t = limb(words[w], l, true);
                t.push_back(Term{P - 1, {var(OUT + 2 * w + l)}});
                b.add(TRANSITION, t);
            }
        // chaining value: equals the working variables where a block starts, constant inside a block
        for (int w = 0; w < 8; w++)
            for (int l = 0; l < 2; l++) {
                Terms t{{1, {var(SEL), var(HC + 2 * w + l)}}};
                append(t, times(limb(words[w], l), var(SEL), true));
                b.add(ALL, t);
            }
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t h = var(HC + i), hn = var(HC + i, true);
            b.add(TRANSITION, Terms{{1, {hn}}, {P - 1, {h}}, {P - 1, {s63, hn}}, {1, {s63, h}}});
        }
        // first row: the IV; last row: the public digest
        for (int w = 0; w < 8; w++)
            for (int l = 0; l < 2; l++) {
                Terms t = limb(words[w], l);
                if (chained) t.push_back(Term{P - 1, {pub(16 + 2 * w + l)}});
                else t.push_back(Term{neg((IV[w] >> (16 * l)) & 0xffffu), {}});
                b.add(FIRST, t);
            }
        for (uint32_t i = 0; i < 16; i++) b.add(LAST, Terms{{1, {var(OUT + i)}}, {P - 1, {pub(i)}}});
        // message schedule window: shifts and the recurrence, both off in round 63 (the next block brings its own 16 words)
        auto gated = [&](const Terms& in) { Terms t = in; append(t, times(in, s63, true)); return t; };
        for (int j = 0; j < 15; j++)
            for (int l = 0; l < 2; l++) {
                Terms t = limb(xw(j), l, true);
                append(t, negated(limb(xw(j + 1), l)));
                b.add(TRANSITION, gated(t));
            }
        for (int l = 0; l < 2; l++) {
            Terms t = limb(Word{false, xl(15)}, l, true);
            for (uint32_t k = 0; k < 2; k++) t.push_back(Term{(uint32_t)(((uint64_t)1 << (16 + k)) % P), {var(CY_SCHED + 2 * l + k)}});
            Terms sum = limb(Word{true, SG1}, l, true);
            append(sum, limb(Word{false, xl(8)}, l, true));
            append(sum, limb(Word{true, SG0}, l, true));
            append(sum, limb(Word{true, X0}, l));
            append(t, negated(sum));
            if (l == 1) for (uint32_t k = 0; k < 2; k++) t.push_back(Term{P - (1u << k), {var(CY_SCHED + k)}});
            b.add(TRANSITION, gated(t));
        }
        // ---- padding (round 5; the header comment states what each group pins).  PB: where the padding's public values start
        {
            const uint32_t PB = chained ? 2 * N_DIGEST : N_DIGEST;
            const uint32_t act = var(ACT), actn = var(ACT, true), cnt = var(CNT), lastb = var(LASTB), l2 = var(L2), sb = var(SB), z0 = var(Z0), z2 = var(Z2), s0 = var(SEL);
            // block count
            b.add(FIRST, Terms{{1, {cnt}}, {P - 1, {pub(PB + PP_K)}}});
            b.add(TRANSITION, Terms{{1, {var(CNT, true)}}, {P - 1, {cnt}}, {1, {s63, act}}});
            b.add(ALL, Terms{{1, {cnt}}, {P - 1, {act, cnt}}});
            b.add(LAST, Terms{{1, {cnt}}, {P - 1, {act}}});
            // the flags: bits, constant inside a block, tied to where ACT drops
            for (uint32_t f : {LASTB, L2}) {
                b.add(ALL, Terms{{1, {var(f), var(f)}}, {P - 1, {var(f)}}});
                b.add(TRANSITION, Terms{{1, {var(f, true)}}, {P - 1, {var(f)}}, {P - 1, {s63, var(f, true)}}, {1, {s63, var(f)}}});
            }
            b.add(TRANSITION, Terms{{1, {s63, act}}, {P - 1, {s63, actn}}, {P - 1, {s63, lastb}}});
            b.add(LAST, Terms{{1, {lastb}}, {P - 1, {act}}});
            b.add(TRANSITION, Terms{{1, {s63, var(LASTB, true)}}, {P - 1, {s63, l2}}});
            b.add(LAST, Terms{{1, {l2}}});
            b.add(ALL, Terms{{1, {z0}}, {P - 1, {s0, lastb}}});
            b.add(ALL, Terms{{1, {z2}}, {P - 1, {s0, l2}}});
            // the row of the boundary word
            {
                Terms t{{1, {sb}}};
                for (uint32_t j = 0; j < 16; j++) { t.push_back(Term{P - 1, {pub(PB + PP_BWL + j), lastb, var(SEL + j)}}); t.push_back(Term{P - 1, {pub(PB + PP_BW2 + j), l2, var(SEL + j)}}); }
                b.add(ALL, t);
            }
            // its bits: with c message bytes in front (KIND_c), bit 31 - 8c is the 1 of 0x80 and everything below it is zero
            for (uint32_t i = 0; i < 32; i++) {
                Terms t;
                for (uint32_t c = 0; c < 4; c++) {
                    if (i <= 31 - 8 * c) t.push_back(Term{1, {pub(PB + PP_KIND + c), sb, var(X0 + i)}});
                    if (i == 31 - 8 * c) t.push_back(Term{P - 1, {pub(PB + PP_KIND + c), sb}});
                }
                b.add(ALL, t);
            }
            // words that must be zero, the length field: at row s_0 of the block, limb by limb
            for (int j = 0; j < 16; j++)
                for (int l = 0; l < 2; l++) {
                    const Terms lj = limb(xw(j), l);
                    Terms t = times(times(lj, z0, false), pub(PB + PP_ZWL + (uint32_t)j), false);
                    append(t, times(times(lj, z2, false), pub(PB + PP_ZW2 + (uint32_t)j), false));
                    if (j <= 13) append(t, times(times(lj, z0, false), pub(PB + PP_Z13), false));
                    else {        // W_14, W_15: the bit length (W_15 low / high limb = LEN_0 / LEN_1, W_14 = LEN_2 / LEN_3) in the trace's last block when the message ends here
                        append(t, times(times(lj, z0, false), pub(PB + PP_FIN), false));
                        t.push_back(Term{P - 1, {pub(PB + PP_FIN), z0, pub(PB + PP_LEN + (j == 15 ? 0u : 2u) + (uint32_t)l)}});
                    }
                    b.add(ALL, t);
                }
        }
        std::vector<uint32_t> p{AIR_MAGIC, 1u, WIDTH, b.count, chained ? N_PUBLIC_CHAINED : N_PUBLIC, (uint32_t)(6 + b.body.size())};
        p.insert(p.end(), b.body.begin(), b.body.end());
        return p;
    }
}
static const std::vector<uint32_t>& program() { static const std::vector<uint32_t> prog = build_program(false); return prog; }
static const std::vector<uint32_t>& program_chained() { static const std::vector<uint32_t> prog = build_program(true); return prog; }

// ---- host side: padding and the chaining values the blocks start from ---------------------------------------------------
static void compress(uint32_t h[8], const uint8_t* block) {
    const uint32_t* K = round_constants().k;
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) | ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
    for (int i = 16; i < 64; i++) w[i] = w[i - 16] + small_sigma0(w[i - 15]) + w[i - 7] + small_sigma1(w[i - 2]);
    uint32_t v[8];
    std::memcpy(v, h, 32);
    for (int r = 0; r < 64; r++) {
        const uint32_t t1 = v[7] + big_sigma1(v[4]) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[r] + w[r];
        const uint32_t t2 = big_sigma0(v[0]) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += v[i];
}

// ---- device side: one workgroup (one wavefront) per block, lane r writes the row of round r ----------------------------------
struct TraceArgs {
    const uint32_t* words;     // [n_blocks][16] message words (big-endian already resolved); inactive blocks: zeros
    const uint32_t* chain;     // [n_blocks][8] chaining value each block starts from
    uint32_t* out;
    uint64_t ld;
    uint32_t active;           // blocks [0, active) belong to the message
    uint32_t pad_block, pad_row; // the block (0xFFFFFFFF: none in this trace) and the row = word index whose W_t holds the 0x80 byte (PadPlace)
    uint32_t k[64];
};

__device__ __forceinline__ uint32_t mbit(uint32_t x, int i) { return ((x >> i) & 1u) ? MONTY_R1 : 0u; }
__device__ __forceinline__ uint32_t mlimb(uint32_t x) { return dmul(x, MONTY_R2); }                    // x < 2^16 -> Montgomery form
__device__ __forceinline__ void put_bits(uint32_t* row, uint32_t base, uint32_t x) {
#pragma unroll
    for (int i = 0; i < 32; i++) row[base + i] = mbit(x, i);
}
__device__ __forceinline__ void put_limbs(uint32_t* row, uint32_t base, uint32_t x) { row[base] = mlimb(x & 0xffffu); row[base + 1] = mlimb(x >> 16); }
// OUT limb pair + carry bits of a sum of up to 8 words: returns nothing, writes columns
__device__ __forceinline__ void put_sum(uint32_t* row, uint32_t out_col, uint32_t cy, uint32_t ncy, const uint32_t* src, int n) {
    uint32_t lo = 0, hi = 0;
    for (int i = 0; i < n; i++) { lo += src[i] & 0xffffu; hi += src[i] >> 16; }
    hi += lo >> 16;
    row[out_col] = mlimb(lo & 0xffffu);
    row[out_col + 1] = mlimb(hi & 0xffffu);
    for (uint32_t q = 0; q < ncy; q++) { row[cy + q] = mbit(lo >> 16, q); row[cy + ncy + q] = mbit(hi >> 16, q); }
}

__device__ __forceinline__ void sha256_trace_kernel_body(const TraceArgs& a) {
    __shared__ uint32_t w[80];
    __shared__ uint32_t st[64][8];
    const uint32_t blk = blockIdx.x, r = threadIdx.x;
    const bool act = blk < a.active;
    if (r < 16) w[r] = a.words[(uint64_t)blk * 16 + r];
    __syncthreads();
    if (r == 0) {
        for (int i = 16; i < 80; i++) w[i] = w[i - 16] + small_sigma0(w[i - 15]) + w[i - 7] + small_sigma1(w[i - 2]);
        uint32_t v[8];
        for (int i = 0; i < 8; i++) v[i] = a.chain[(uint64_t)blk * 8 + i];
        for (int t = 0; t < 64; t++) {
            for (int i = 0; i < 8; i++) st[t][i] = v[i];
            const uint32_t t1 = v[7] + big_sigma1(v[4]) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + a.k[t] + w[t];
            const uint32_t t2 = big_sigma0(v[0]) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
        }
    }
    __syncthreads();
    uint32_t* row = a.out + ((uint64_t)blk * 64 + r) * a.ld;
    const uint32_t va = st[r][0], vb = st[r][1], vc = st[r][2], vd = st[r][3], ve = st[r][4], vf = st[r][5], vg = st[r][6], vh = st[r][7];
    for (uint32_t t = 0; t < 64; t++) row[SEL + t] = t == r ? MONTY_R1 : 0u;
    put_bits(row, A, va); put_bits(row, B, vb); put_bits(row, C, vc);
    put_bits(row, E, ve); put_bits(row, F, vf); put_bits(row, G, vg);
    put_limbs(row, D, vd); put_limbs(row, HV, vh);
    const uint32_t s1 = big_sigma1(ve), ch = (ve & vf) ^ (~ve & vg), s0 = big_sigma0(va), mj = (va & vb) ^ (va & vc) ^ (vb & vc);
    put_bits(row, S1, s1); put_bits(row, CH, ch); put_bits(row, S0, s0); put_bits(row, MJ, mj);
    uint32_t hc[8];
    for (int i = 0; i < 8; i++) { hc[i] = a.chain[(uint64_t)blk * 8 + i]; put_limbs(row, HC + 2 * i, hc[i]); }
    const bool last = r == 63, skip = last && !act;
    const uint32_t cy[8] = {CY_A, CY_W6, CY_W6 + 2, CY_W6 + 4, CY_E, CY_W6 + 6, CY_W6 + 8, CY_W6 + 10};
    for (int i = 0; i < 8; i++) {
        uint32_t src[8];
        int n = 0;
        if (!skip) {
            if (i == 0) { src[n++] = vh; src[n++] = s1; src[n++] = ch; src[n++] = a.k[r]; src[n++] = w[r]; src[n++] = s0; src[n++] = mj; }
            else if (i == 4) { src[n++] = vd; src[n++] = vh; src[n++] = s1; src[n++] = ch; src[n++] = a.k[r]; src[n++] = w[r]; }
            else src[n++] = st[r][i - 1];
        }
        if (last) src[n++] = hc[i];
        put_sum(row, OUT + 2 * i, cy[i], (i == 0 || i == 4) ? 3u : 1u, src, n);
    }
    put_bits(row, X0, w[r]); put_bits(row, X13, w[r + 13]);
    for (int j = 1; j < 16; j++) if (j != 13) put_limbs(row, xl(j), w[r + j]);
    put_bits(row, SG0, small_sigma0(w[r])); put_bits(row, SG1, small_sigma1(w[r + 13]));
    {
        uint32_t lo = 0, hi = 0;
        if (!last) {
            const uint32_t parts[4] = {small_sigma1(w[r + 14]), w[r + 9], small_sigma0(w[r + 1]), w[r]};
            for (int i = 0; i < 4; i++) { lo += parts[i] & 0xffffu; hi += parts[i] >> 16; }
            hi += lo >> 16;
        }
        for (uint32_t q = 0; q < 2; q++) { row[CY_SCHED + q] = mbit(lo >> 16, q); row[CY_SCHED + 2 + q] = mbit(hi >> 16, q); }
    }
    row[ACT] = act ? MONTY_R1 : 0u;
    row[SKIP] = skip ? MONTY_R1 : 0u;
    // the padding columns: blocks left, the last active block and the one before it, the boundary word's row, the two s_0 products
    const bool lastb = blk + 1 == a.active, l2 = blk + 2 == a.active;
    row[CNT] = act ? mlimb(a.active - blk) : 0u;
    row[LASTB] = lastb ? MONTY_R1 : 0u;
    row[L2] = l2 ? MONTY_R1 : 0u;
    row[SB] = (blk == a.pad_block && r == a.pad_row) ? MONTY_R1 : 0u;
    row[Z0] = (r == 0 && lastb) ? MONTY_R1 : 0u;
    row[Z2] = (r == 0 && l2) ? MONTY_R1 : 0u;
    for (uint32_t c = USED; c < WIDTH; c++) row[c] = 0u;              // the unused columns of the last tile
}
__global__ void __launch_bounds__(64) sha256_trace_kernel(TraceArgs a) { sha256_trace_kernel_body(a); }
struct sha256_trace_kernel_bargs { TraceArgs a; static sha256_trace_kernel_bargs make(TraceArgs a) { return sha256_trace_kernel_bargs{a}; } };
__global__ void __launch_bounds__(64) sha256_trace_kernel_batch(const sha256_trace_kernel_bargs* __restrict__ zk_arr) { const sha256_trace_kernel_bargs& zk_b = zk_arr[blockIdx.z]; sha256_trace_kernel_body(zk_b.a); }



// ---- multiplicities of a range table on the device: count how often every value of [0, 2^log_table) appears in the listed columns
// of a trace (Montgomery words), then write the table's two columns: value v = row index, multiplicity m = the count
struct HistArgs { const uint32_t* trace; uint64_t ld; uint64_t rows; uint32_t cols[16]; uint32_t n_cols; uint32_t log_table; uint32_t* counts; uint32_t* bad; };
__device__ __forceinline__ void lookup_hist_kernel_body(const HistArgs& a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows) return;
    const uint32_t* row = a.trace + i * a.ld;
    for (uint32_t k = 0; k < a.n_cols; k++) {
        const uint32_t v = from_monty(row[a.cols[k]]);
        if (v >> a.log_table) atomicAdd(a.bad, 1u);          // a value the table does not hold: the caller hears about it
        else atomicAdd(a.counts + v, 1u);
    }
}
__global__ void __launch_bounds__(256) lookup_hist_kernel(HistArgs a) { lookup_hist_kernel_body(a); }
struct lookup_hist_kernel_bargs { HistArgs a; static lookup_hist_kernel_bargs make(HistArgs a) { return lookup_hist_kernel_bargs{a}; } };
__global__ void __launch_bounds__(256) lookup_hist_kernel_batch(const lookup_hist_kernel_bargs* __restrict__ zk_arr) { const lookup_hist_kernel_bargs& zk_b = zk_arr[blockIdx.z]; lookup_hist_kernel_body(zk_b.a); }

__device__ __forceinline__ void range_table_kernel_body(const uint32_t* counts, uint32_t rows, uint32_t* out, uint64_t ld, uint32_t value_col, uint32_t mult_col) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    out[(uint64_t)i * ld + value_col] = dmul(i, MONTY_R2);
    out[(uint64_t)i * ld + mult_col] = dmul(counts[i] % P, MONTY_R2);
}
__global__ void __launch_bounds__(256) range_table_kernel(const uint32_t* counts, uint32_t rows, uint32_t* out, uint64_t ld, uint32_t value_col, uint32_t mult_col) { range_table_kernel_body(counts, rows, out, ld, value_col, mult_col); }
struct range_table_kernel_bargs { const uint32_t* counts; uint32_t rows; uint32_t* out; uint64_t ld; uint32_t value_col; uint32_t mult_col; static range_table_kernel_bargs make(const uint32_t* counts, uint32_t rows, uint32_t* out, uint64_t ld, uint32_t value_col, uint32_t mult_col) { return range_table_kernel_bargs{counts, rows, out, ld, value_col, mult_col}; } };
__global__ void __launch_bounds__(256) range_table_kernel_batch(const range_table_kernel_bargs* __restrict__ zk_arr) { const range_table_kernel_bargs& zk_b = zk_arr[blockIdx.z]; range_table_kernel_body(zk_b.counts, zk_b.rows, zk_b.out, zk_b.ld, zk_b.value_col, zk_b.mult_col); }


}  // namespace sha
}  // namespace zk

using namespace zk;

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

static int log2_exact(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return ((size_t)1 << l) == n ? l : -1; }

extern "C" {

int zkhip_range_table(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, size_t rows, const uint32_t* columns, int n_columns, int log_table,
                      uint32_t* d_table, size_t table_ld, uint32_t value_col, uint32_t mult_col) {
    CHECK_CTX(ctx);
    if (!d_trace || !columns || !d_table || n_columns < 1 || n_columns > 16 || log_table < 5 || log_table > 22 || rows == 0 ||
        value_col >= table_ld || mult_col >= table_ld || value_col == mult_col)
        return fail(ZKHIP_ERR_INVALID, "range_table: 1..16 columns, log_table in [5, 22], distinct value / multiplicity columns inside the row pitch");
    const size_t n = (size_t)1 << log_table;
    void* v_counts;
    ZK_TRY(ctx_reserve(ctx, S_ADDEND, (n + 1) * 4, &v_counts));
    ZK_TRY(dev_memset(ctx, v_counts, 0, (n + 1) * 4));
    sha::HistArgs a{};
    a.trace = d_trace; a.ld = ld; a.rows = rows; a.n_cols = (uint32_t)n_columns; a.log_table = (uint32_t)log_table;
    for (int k = 0; k < n_columns; k++) { if (columns[k] >= ld) return fail(ZKHIP_ERR_INVALID, "range_table: column outside the row pitch"); a.cols[k] = columns[k]; }
    a.counts = (uint32_t*)v_counts; a.bad = (uint32_t*)v_counts + n;
    ZK_LAUNCH(sha::lookup_hist_kernel, sha::lookup_hist_kernel_batch, sha::lookup_hist_kernel_bargs, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, a);
    ZK_HIP(hipGetLastError());
    ZK_LAUNCH(sha::range_table_kernel, sha::range_table_kernel_batch, sha::range_table_kernel_bargs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)v_counts, (uint32_t)n, d_table, (uint64_t)table_ld, value_col, mult_col);
    ZK_HIP(hipGetLastError());
    uint32_t bad = 0;
    ZK_TRY(dev_d2h(ctx, &bad, a.bad, 4));
    if (bad) return fail(ZKHIP_ERR_INVALID, "range_table: " + std::to_string(bad) + " looked-up values lie outside [0, 2^log_table)");
    return ZKHIP_OK;
}

size_t zkhip_sha256_air(uint32_t* program, size_t cap_words) {
    const std::vector<uint32_t>& p = sha::program();
    if (program && cap_words >= p.size()) std::memcpy(program, p.data(), p.size() * 4);
    return p.size();
}

size_t zkhip_sha256_pad(const uint8_t* message, size_t len, uint8_t* blocks, size_t cap) {
    const size_t padded = ((len + 9 + 63) / 64) * 64;
    if (!blocks || cap < padded || (len && !message)) return padded;
    if (len) std::memcpy(blocks, message, len);
    std::memset(blocks + len, 0, padded - len);
    blocks[len] = 0x80;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) blocks[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
    return padded;
}

void zkhip_sha256_digest(const uint8_t* message, size_t len, uint8_t digest[32]) {
    std::vector<uint8_t> blocks(((len + 9 + 63) / 64) * 64);
    zkhip_sha256_pad(message, len, blocks.data(), blocks.size());
    uint32_t h[8];
    std::memcpy(h, sha::IV, 32);
    for (size_t k = 0; k < blocks.size() / 64; k++) sha::compress(h, blocks.data() + 64 * k);
    for (int i = 0; i < 8; i++) { digest[4 * i] = (uint8_t)(h[i] >> 24); digest[4 * i + 1] = (uint8_t)(h[i] >> 16); digest[4 * i + 2] = (uint8_t)(h[i] >> 8); digest[4 * i + 3] = (uint8_t)h[i]; }
}

void zkhip_sha256_padding_publics(uint64_t message_len, uint64_t first_block, uint64_t n_active, uint32_t out[75]) {
    static_assert(sha::N_PAD == 75, "the header says 75");
    (void)sha::padding_publics(message_len, first_block, n_active, out);
}

int zkhip_sha256_gen_trace(zkhip_ctx* ctx, const uint8_t* blocks, size_t n_active, size_t n_blocks, uint64_t message_len, uint32_t* d_trace, size_t ld,
                           uint32_t publics[91]) {
    if (n_active != (size_t)((message_len + 8) / 64 + 1)) return fail(ZKHIP_ERR_INVALID, "sha256_gen_trace: n_active must be the padded message's block count (message_len + 8) / 64 + 1");
    return zkhip_sha256_gen_trace_chained(ctx, sha::IV, blocks, n_active, n_blocks, message_len, 0, d_trace, ld, publics);
}
// publics: [0, 16) the final chaining value's limbs, then the 75 padding values of this slice (the caller of the chained program puts the
// initial value's 16 limbs between them)
int zkhip_sha256_gen_trace_chained(zkhip_ctx* ctx, const uint32_t chain_in[8], const uint8_t* blocks, size_t n_active, size_t n_blocks, uint64_t message_len,
                                   uint64_t first_block, uint32_t* d_trace, size_t ld, uint32_t publics[91]) {
    CHECK_CTX(ctx);
    if (!chain_in) return fail(ZKHIP_ERR_INVALID, "sha256_gen_trace: null chaining value");
    const int lb = log2_exact(n_blocks);
    if (!blocks || !d_trace || !publics || n_active == 0 || n_active > n_blocks || lb < 0 || lb + 6 > MAX_LOG_ROWS || ld < sha::WIDTH)
        return fail(ZKHIP_ERR_INVALID, "sha256_gen_trace: 1 <= n_active <= n_blocks = 2^k <= 2^16, ld >= 640");
    if (first_block + n_active > (message_len + 8) / 64 + 1) return fail(ZKHIP_ERR_INVALID, "sha256_gen_trace: the slice runs past the padded message");
    std::vector<uint32_t> host((size_t)n_blocks * 24, 0u);             // [n_blocks][16] words, then [n_blocks][8] chaining values
    uint32_t* words = host.data();
    uint32_t* chain = host.data() + n_blocks * 16;
    uint32_t h[8];
    std::memcpy(h, chain_in, 32);
    for (size_t k = 0; k < n_blocks; k++) {
        std::memcpy(chain + 8 * k, h, 32);
        if (k < n_active) {
            const uint8_t* b = blocks + 64 * k;
            for (int i = 0; i < 16; i++) words[16 * k + i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
            sha::compress(h, b);
        }
    }
    for (int i = 0; i < 8; i++) { publics[2 * i] = h[i] & 0xffffu; publics[2 * i + 1] = h[i] >> 16; }
    const sha::PadPlace pl = sha::padding_publics(message_len, first_block, n_active, publics + sha::N_DIGEST);
    void* stage;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, host.size() * 4, &stage));
    ZK_TRY(dev_h2d(ctx, stage, host.data(), host.size() * 4));           // (returns when `host` may go out of scope)
    sha::TraceArgs a;
    a.words = (const uint32_t*)stage;
    a.chain = (const uint32_t*)stage + n_blocks * 16;
    a.out = d_trace;
    a.ld = ld;
    a.active = (uint32_t)n_active;
    a.pad_block = pl.pad_block; a.pad_row = pl.pad_row;
    std::memcpy(a.k, sha::round_constants().k, sizeof(a.k));
    ZK_LAUNCH(sha::sha256_trace_kernel, sha::sha256_trace_kernel_batch, sha::sha256_trace_kernel_bargs, dim3((unsigned)n_blocks), dim3(64), 0, ctx->stream, a);
    ZK_HIP(hipGetLastError());
    return ZKHIP_OK;
}

static int sha_shape(size_t message_len, size_t* padded, size_t* n_active, size_t* n_blocks, int* log_n) {
    *padded = ((message_len + 9 + 63) / 64) * 64;
    *n_active = *padded / 64;
    size_t nb = 1;
    while (nb < *n_active) nb <<= 1;
    *n_blocks = nb;
    *log_n = 6 + log2_exact(nb);
    return *log_n <= MAX_LOG_ROWS ? ZKHIP_OK : fail(ZKHIP_ERR_INVALID, "sha256: message longer than 2^16 blocks");
}

size_t zkhip_sha256_proof_size(size_t message_len, const zkhip_params* prm) {
    size_t padded, na, nb;
    int log_n;
    if (sha_shape(message_len, &padded, &na, &nb, &log_n) != ZKHIP_OK) return 0;
    const std::vector<uint32_t>& p = sha::program();
    return zkhip_proof_size_air(p.data(), p.size(), log_n, sha::WIDTH, prm, sha::N_PUBLIC);
}

int zkhip_prove_sha256(zkhip_ctx* ctx, const uint8_t* message, size_t message_len, const zkhip_params* prm, uint8_t digest[32],
                       uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if ((message_len && !message) || !digest || !proof || !len || !prm) return fail(ZKHIP_ERR_INVALID, "prove_sha256: null argument");
    size_t padded, na, nb;
    int log_n;
    ZK_TRY(sha_shape(message_len, &padded, &na, &nb, &log_n));
    std::vector<uint8_t> blocks(padded);
    zkhip_sha256_pad(message, message_len, blocks.data(), padded);
    void* trace;
    ZK_TRY(ctx_reserve(ctx, S_CHIP, ((size_t)sha::WIDTH << log_n) * 4, &trace));
    uint32_t limbs[sha::N_PUBLIC];
    ZK_TRY(zkhip_sha256_gen_trace(ctx, blocks.data(), na, nb, message_len, (uint32_t*)trace, sha::WIDTH, limbs));
    for (int i = 0; i < 8; i++) {
        const uint32_t w = limbs[2 * i] | (limbs[2 * i + 1] << 16);
        digest[4 * i] = (uint8_t)(w >> 24); digest[4 * i + 1] = (uint8_t)(w >> 16); digest[4 * i + 2] = (uint8_t)(w >> 8); digest[4 * i + 3] = (uint8_t)w;
    }
    const std::vector<uint32_t>& p = sha::program();
    return zkhip_prove_shard_air(ctx, p.data(), p.size(), (const uint32_t*)trace, sha::WIDTH, log_n, sha::WIDTH, limbs, sha::N_PUBLIC, prm, proof, cap, len);
}

// the public values of "digest = SHA-256(a message of message_len bytes)": the digest's limbs, then the padding's
static void sha_statement(const uint8_t digest[32], uint64_t message_len, uint32_t pv[sha::N_PUBLIC]) {
    for (int i = 0; i < 8; i++) {
        const uint32_t w = ((uint32_t)digest[4 * i] << 24) | ((uint32_t)digest[4 * i + 1] << 16) | ((uint32_t)digest[4 * i + 2] << 8) | digest[4 * i + 3];
        pv[2 * i] = w & 0xffffu; pv[2 * i + 1] = w >> 16;
    }
    (void)sha::padding_publics(message_len, 0, (message_len + 8) / 64 + 1, pv + sha::N_DIGEST);
}
int zkhip_verify_sha256(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const zkhip_params* prm, int* reason) {
    if (!proof || !digest || !prm || len < 16) return fail(ZKHIP_ERR_INVALID, "verify_sha256: null argument");
    uint32_t head[4];
    std::memcpy(head, proof, 16);
    const int log_n = (int)head[2];                                    // the trace height is read from the proof and bound by its transcript; the block COUNT is the statement's
    if (log_n < 6 || log_n > MAX_LOG_ROWS || (message_len + 8) / 64 + 1 > ((uint64_t)1 << (log_n - 6))) {
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_sha256: not a SHA-256 chip proof for a message of this length");
    }
    uint32_t limbs[sha::N_PUBLIC];
    sha_statement(digest, message_len, limbs);
    const std::vector<uint32_t>& p = sha::program();
    return zkhip_verify_shard_air(p.data(), p.size(), proof, len, log_n, sha::WIDTH, limbs, sha::N_PUBLIC, prm, reason);
}

// ---- the SHA-256 guest as a keyed machine: setup once, then one proof per message ------------------------------------------
// The compression chip's own constraints range-check every limb except four (OUT of d and h: they are only added, never
// decomposed).  In the machine form those four go to a 2^16-row range table, as SP1's chips send their limbs to the byte / range
// tables: the table's VALUES are a preprocessed column (fixed by the key, no counter constraints), its multiplicities a main column.
// setup = the reference's client.setup (sp1.rs:113): commits the table once; the root is the verifying key.
}  // extern "C"
namespace zk {
namespace sha {
constexpr uint32_t RANGE_BUS = 16, RANGE_LOG = 16;
static const uint32_t SENT[4] = {OUT + 6, OUT + 7, OUT + 14, OUT + 15};
static const std::vector<uint32_t>& range_program() {          // combined row [v 0 0 0 | v m 0 0]: one harmless first-row identity
    static const std::vector<uint32_t> p{AIR_MAGIC, 1u, 8u, 1u, N_PUBLIC, 6u + 5u, FIRST, 1u, 1u, 1u, var(0)};
    return p;
}
static const std::vector<uint32_t>& sha_interactions() {
    static const std::vector<uint32_t> t = [] {
        std::vector<uint32_t> v{LOOKUP_MAGIC, 4u, 3u + 4u * 5u};
        for (uint32_t c : SENT) { v.push_back(0u); v.push_back(0xFFFFFFFFu); v.push_back(RANGE_BUS); v.push_back(1u); v.push_back(c); }
        return v;
    }();
    return t;
}
static const std::vector<uint32_t>& range_interactions() {     // receive (multiplicity = combined column 5, [preprocessed value])
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 1u, 3u + 5u, 1u, 5u, RANGE_BUS, 1u, 0u};
    return t;
}
// the machine of a message: chips tallest first -- the chip before the table once it is taller than 2^16 rows
struct MachineShape {
    int n = 2, sha_at, table_at;
    int32_t log_ns[2]; uint32_t widths[2], pre_widths[2]; int32_t entries[2];
    const uint32_t* progs[2]; size_t prog_words[2]; const uint32_t* tabs[2]; size_t tab_words[2];
};
static MachineShape machine_shape(int log_n) {
    MachineShape m;
    m.sha_at = log_n > (int)RANGE_LOG ? 0 : 1; m.table_at = 1 - m.sha_at;
    m.log_ns[m.sha_at] = log_n; m.widths[m.sha_at] = WIDTH; m.pre_widths[m.sha_at] = 0; m.entries[m.sha_at] = -1;
    m.progs[m.sha_at] = program().data(); m.prog_words[m.sha_at] = program().size();
    m.tabs[m.sha_at] = sha_interactions().data(); m.tab_words[m.sha_at] = sha_interactions().size();
    m.log_ns[m.table_at] = (int)RANGE_LOG; m.widths[m.table_at] = 4; m.pre_widths[m.table_at] = 4; m.entries[m.table_at] = 0;
    m.progs[m.table_at] = range_program().data(); m.prog_words[m.table_at] = range_program().size();
    m.tabs[m.table_at] = range_interactions().data(); m.tab_words[m.table_at] = range_interactions().size();
    return m;
}
}  // namespace sha
}  // namespace zk
// lock-step batches of small transcripts (batch.h): the tallest chip that still counts as small (2^14 rows x 640 columns = 10 M cells, a
// 16 KB transcript -- the size measured as launch-bound; context.h LOCKSTEP_MAX_CELLS)
constexpr int LOCKSTEP_MAX_LOG_N = 14;
namespace zk { extern std::atomic<uint64_t> g_lockstep_stats[6]; }
extern "C" {

int zkhip_sha256_setup(zkhip_ctx* ctx, const zkhip_params* prm, zkhip_machine_key** key, uint32_t vk[8]) {
    CHECK_CTX(ctx);
    if (!prm || !key || !vk) return fail(ZKHIP_ERR_INVALID, "sha256_setup: null argument");
    const size_t n = (size_t)1 << sha::RANGE_LOG;
    std::vector<uint32_t> values(n * 4, 0u);
    for (size_t v = 0; v < n; v++) values[4 * v] = to_monty((uint32_t)v);
    void* d;
    ZK_TRY(ctx_reserve(ctx, S_CHIP_B, n * 16, &d));
    ZK_HIP(hipMemcpyAsync(d, values.data(), n * 16, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    zkhip_chip pre{};
    pre.d_trace = (const uint32_t*)d; pre.ld = 4; pre.log_n = (int)sha::RANGE_LOG; pre.width = 4; pre.logup_pairs = 0; pre.partner = -1;
    return zkhip_machine_setup(ctx, &pre, 1, prm, key, vk);
}

size_t zkhip_sha256_machine_proof_size(size_t message_len, const zkhip_params* prm) {
    size_t padded, na, nb;
    int log_n;
    if (sha_shape(message_len, &padded, &na, &nb, &log_n) != ZKHIP_OK || log_n > 20) return 0;
    const sha::MachineShape m = sha::machine_shape(log_n);
    return zkhip_machine_proof_size_keyed(m.log_ns, m.widths, m.pre_widths, m.progs, m.prog_words, m.tabs, m.tab_words, 2, prm, sha::N_PUBLIC);
}

int zkhip_prove_sha256_machine(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint8_t* message, size_t message_len, const zkhip_params* prm,
                               uint8_t digest[32], uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!key || (message_len && !message) || !digest || !proof || !len || !prm) return fail(ZKHIP_ERR_INVALID, "prove_sha256_machine: null argument");
    size_t padded, na, nb;
    int log_n;
    ZK_TRY(sha_shape(message_len, &padded, &na, &nb, &log_n));
    if (log_n > 20) return fail(ZKHIP_ERR_INVALID, "prove_sha256_machine: the machine prover takes chips of up to 2^20 rows (2^14 blocks, 1 MiB)");
    std::vector<uint8_t> blocks(padded);
    zkhip_sha256_pad(message, message_len, blocks.data(), padded);
    void *trace, *counts;
    ZK_TRY(ctx_reserve(ctx, S_CHIP, ((size_t)sha::WIDTH << log_n) * 4, &trace));
    ZK_TRY(ctx_reserve(ctx, S_CHIP_B, ((size_t)1 << sha::RANGE_LOG) * 16, &counts));
    uint32_t limbs[sha::N_PUBLIC];
    ZK_TRY(zkhip_sha256_gen_trace(ctx, blocks.data(), na, nb, message_len, (uint32_t*)trace, sha::WIDTH, limbs));
    for (int i = 0; i < 8; i++) {
        const uint32_t w = limbs[2 * i] | (limbs[2 * i + 1] << 16);
        digest[4 * i] = (uint8_t)(w >> 24); digest[4 * i + 1] = (uint8_t)(w >> 16); digest[4 * i + 2] = (uint8_t)(w >> 8); digest[4 * i + 3] = (uint8_t)w;
    }
    // the table's main columns (v, multiplicity, 0, 0), counted on the device
    ZK_TRY(dev_memset(ctx, counts, 0, ((size_t)1 << sha::RANGE_LOG) * 16));
    ZK_TRY(zkhip_range_table(ctx, (const uint32_t*)trace, sha::WIDTH, (size_t)1 << log_n, sha::SENT, 4, (int)sha::RANGE_LOG, (uint32_t*)counts, 4, 0, 1));
    const sha::MachineShape m = sha::machine_shape(log_n);
    zkhip_chip chips[2]{};
    chips[m.sha_at].d_trace = (const uint32_t*)trace; chips[m.sha_at].ld = sha::WIDTH; chips[m.sha_at].log_n = log_n; chips[m.sha_at].width = sha::WIDTH;
    chips[m.table_at].d_trace = (const uint32_t*)counts; chips[m.table_at].ld = 4; chips[m.table_at].log_n = (int)sha::RANGE_LOG; chips[m.table_at].width = 4;
    chips[0].partner = chips[1].partner = -1;
    return zkhip_prove_machine_keyed_at(ctx, key, m.entries, chips, m.progs, m.prog_words, m.tabs, m.tab_words, 2, limbs, sha::N_PUBLIC, prm, proof, cap, len);
}

// the keyed machine of a message of this length as data: chip `which` (0, 1: tallest first), kind 0 its program, 1 its interaction table -- what a
// zkhip_machine_desc needs to hand the machine's proofs to zkhip_prove_machine_verifier (64 transcript proofs -> one)
size_t zkhip_sha256_machine_describe(size_t message_len, int which, int kind, uint32_t* out, size_t cap, int* log_n, uint32_t* width, uint32_t* pre_width) {
    size_t padded, na, nb;
    int ln;
    if (sha_shape(message_len, &padded, &na, &nb, &ln) != ZKHIP_OK || ln > 20 || which < 0 || which > 1 || kind < 0 || kind > 1) return 0;
    const sha::MachineShape m = sha::machine_shape(ln);
    if (log_n) *log_n = m.log_ns[which];
    if (width) *width = m.widths[which];
    if (pre_width) *pre_width = m.pre_widths[which];
    const uint32_t* src = kind ? m.tabs[which] : m.progs[which];
    const size_t n = kind ? m.tab_words[which] : m.prog_words[which];
    if (out && cap >= n) std::memcpy(out, src, n * 4);
    return n;
}

int zkhip_verify_sha256_machine(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const uint32_t vk[8], const zkhip_params* prm, int* reason) {
    if (!proof || !digest || !vk || !prm || len < 4 * 18) return fail(ZKHIP_ERR_INVALID, "verify_sha256_machine: null argument");
    uint32_t head[18];
    std::memcpy(head, proof, sizeof head);
    // two header entries (log_n, width, has-program, interactions, preprocessed width): the chip's height is read from the proof
    // and bound by its transcript; everything else about the machine is fixed here
    const int sha_at = head[8 + 1] == sha::WIDTH ? 0 : 1;
    const int log_n = (int)head[8 + 5 * sha_at];
    if (head[2] != 2u || log_n < 6 || log_n > 20 || (sha_at == 0) != (log_n > (int)sha::RANGE_LOG) || (message_len + 8) / 64 + 1 > ((uint64_t)1 << (log_n - 6))) {
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_sha256_machine: not a proof of the SHA-256 machine for a message of this length");
    }
    uint32_t limbs[sha::N_PUBLIC];
    sha_statement(digest, message_len, limbs);
    const sha::MachineShape m = sha::machine_shape(log_n);
    return zkhip_verify_machine_keyed(proof, len, m.log_ns, m.widths, m.pre_widths, vk, m.progs, m.prog_words, m.tabs, m.tab_words, 2, limbs, sha::N_PUBLIC, prm, reason);
}

// A batch of transcripts in ONE call (BASELINE configs[2]: sixty-four independent TLS transcripts, shard-parallel over the GPUs): job i is
// proven on devices[i mod n_devices], `in_flight_per_device` at a time on each, every worker on a pooled context that keeps its own
// proving key (setup runs once per context and proof shape; every context arrives at the same vk).  Per job: padding, trace generation
// and the range table's multiplicities on the device, the keyed machine's proof into the job's host buffer, the digest.
// (keyed: the keyed SHA-256 machine, version 11; otherwise the chip alone as zkhip_prove_sha256 makes it, version 7 -- vk unused)
static int prove_transcripts_impl(const int* devices, int n_devices, zkhip_transcript_job* jobs, int n_jobs, const zkhip_params* prm,
                                  int in_flight_per_device, int verify, uint32_t vk[8], bool keyed) {
    uint32_t vk_unused[8];
    if (!keyed) vk = vk_unused;
    if (!jobs || n_jobs < 0 || !prm || !vk) return fail(ZKHIP_ERR_INVALID, "prove_transcripts: bad arguments");
    for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_INVALID; jobs[i].proof_len = 0; }
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_transcripts", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) jobs[i].status = ZKHIP_ERR_NO_DEVICE;
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    std::mutex mu;
    bool have_vk = false;
    std::memset(vk, 0, 32);
    std::vector<char> ran;
    std::unique_ptr<HostPool> checkers;                          // made below when lock-step lanes run with verify
    std::vector<std::string> check_msg((size_t)n_jobs);
    auto run = [&](zkhip_ctx* ctx, int i) {
        zkhip_transcript_job& j = jobs[i];
        int r = ZKHIP_OK;
        if (keyed) {
            if (ctx->sha_key && ctx->sha_key_blowup != prm->log_blowup) { zkhip_machine_key_destroy(ctx->sha_key); ctx->sha_key = nullptr; }
            if (!ctx->sha_key) {
                r = zkhip_sha256_setup(ctx, prm, &ctx->sha_key, ctx->sha_vk);
                ctx->sha_key_blowup = prm->log_blowup;
            }
            if (r == ZKHIP_OK) {
                std::lock_guard<std::mutex> lk(mu);
                if (!have_vk) { std::memcpy(vk, ctx->sha_vk, 32); have_vk = true; }
                else if (std::memcmp(vk, ctx->sha_vk, 32) != 0) r = fail(ZKHIP_ERR_INTERNAL, "prove_transcripts: two contexts disagree about the verifying key");
            }
        }
        size_t len = 0;
        if (r == ZKHIP_OK) r = keyed ? zkhip_prove_sha256_machine(ctx, ctx->sha_key, j.message, j.message_len, prm, j.digest, j.proof, j.proof_cap, &len)
                                     : zkhip_prove_sha256(ctx, j.message, j.message_len, prm, j.digest, j.proof, j.proof_cap, &len);
        batch_leave();                                           // (lock-step batch: the rest is host work)
        // the reference checks every proof right after proving it (sp1.rs:120): on a host thread, while the GPU runs the other proofs --
        // this worker's own thread, or (lock-step lanes, whose members share one thread) a small pool beside the lanes
        j.status = r;
        j.proof_len = r == ZKHIP_OK ? len : 0;
        if (r == ZKHIP_OK && verify) {
            if (checkers && t_batcher) {
                uint32_t key[8];
                std::memcpy(key, ctx->sha_vk, 32);
                const zkhip_params p = *prm;
                zkhip_transcript_job* jp = &j;
                std::string* msg = &check_msg[(size_t)i];
                checkers->submit([jp, len, p, msg, key, keyed] {
                    const int v = keyed ? zkhip_verify_sha256_machine(jp->proof, len, jp->digest, jp->message_len, key, &p, nullptr)
                                        : zkhip_verify_sha256(jp->proof, len, jp->digest, jp->message_len, &p, nullptr);
                    if (v != ZKHIP_OK) { jp->status = v; jp->proof_len = 0; *msg = zkhip_last_error(); }
                });
            } else {
                r = keyed ? zkhip_verify_sha256_machine(j.proof, len, j.digest, j.message_len, ctx->sha_vk, prm, nullptr)
                          : zkhip_verify_sha256(j.proof, len, j.digest, j.message_len, prm, nullptr);
                j.status = r;
                j.proof_len = r == ZKHIP_OK ? len : 0;
            }
        }
        return r;
    };
    // Small transcripts are launch-bound (a few hundred kernels of microseconds each): those of one trace height are proven in
    // lock-step batches whose kernel launches merge (batch.h); the others -- and everything when lock-step is switched off -- are
    // dealt one context, one stream each.  The proofs are the same bytes either way.
    const int max_batch = lockstep_batch();
    std::vector<int> small, big, shape;
    for (int i = 0; i < n_jobs; i++) {
        size_t padded, na, nb;
        int log_n = 0;
        const bool ok = sha_shape(jobs[i].message_len, &padded, &na, &nb, &log_n) == ZKHIP_OK;
        if (ok && max_batch > 1 && log_n <= LOCKSTEP_MAX_LOG_N) { small.push_back(i); shape.push_back(log_n); }
        else big.push_back(i);
    }
    const int nd = (int)devs.size();
    if ((int)small.size() < 2 * nd) { big.insert(big.end(), small.begin(), small.end()); std::sort(big.begin(), big.end()); small.clear(); }
    int rc_small = ZKHIP_OK, rc_big = ZKHIP_OK;
    std::string msg_small;
    if (!small.empty()) {
        if (verify) checkers.reset(new HostPool(8));
        int tallest = 0;
        for (int ln : shape) if (ln > tallest) tallest = ln;
        const uint64_t cells = (((uint64_t)sha::WIDTH << tallest) + ((uint64_t)8 << 16)) << (prm->log_blowup > 1 ? prm->log_blowup - 1 : 0);
        rc_small = deal_jobs_lockstep(devs.data(), nd, (int)small.size(), shape.data(), max_batch, lockstep_lanes(),
                                      [&](zkhip_ctx* ctx, int k) { return run(ctx, small[(size_t)k]); }, ran, cells);
        if (rc_small != ZKHIP_OK) msg_small = zkhip_last_error();
    }
    if (!big.empty())
        rc_big = deal_jobs(devs.data(), nd, (int)big.size(), in_flight_per_device, [&](zkhip_ctx* ctx, int k) { return run(ctx, big[(size_t)k]); }, ran);
    if (checkers) {                                             // the checks made beside the lanes: the lowest rejected job speaks
        checkers->wait();
        for (int i = 0; i < n_jobs && rc_small == ZKHIP_OK; i++)
            if (!check_msg[(size_t)i].empty()) { rc_small = jobs[i].status; msg_small = check_msg[(size_t)i]; }
    }
    if (rc_small != ZKHIP_OK && (rc_big == ZKHIP_OK || small[0] < big[0])) { set_error(msg_small); return rc_small; }
    return rc_big;
}
int zkhip_prove_transcripts(const int* devices, int n_devices, zkhip_transcript_job* jobs, int n_jobs, const zkhip_params* prm,
                            int in_flight_per_device, int verify, uint32_t vk[8]) {
    return prove_transcripts_impl(devices, n_devices, jobs, n_jobs, prm, in_flight_per_device, verify, vk, true);
}
int zkhip_prove_transcripts_air(const int* devices, int n_devices, zkhip_transcript_job* jobs, int n_jobs, const zkhip_params* prm,
                                int in_flight_per_device, int verify) {
    return prove_transcripts_impl(devices, n_devices, jobs, n_jobs, prm, in_flight_per_device, verify, nullptr, false);
}

void zkhip_set_lockstep(int max_batch, int lanes) { lockstep_set(max_batch, lanes); }
int zkhip_selftest_lockstep(int members, int rounds) {
    const int rc = lockstep_selftest(members, rounds);
    return rc ? rc : lockstep_selftest_observe(members);
}
void zkhip_lockstep_stats(uint64_t out[6]) {
    for (int i = 0; i < 6; i++) out[i] = g_lockstep_stats[i].load();
}
uint64_t zkhip_lockstep_stack_high_water(void) { return lockstep_stack_high_water(); }

// ---- a message of ANY length as a chain of shards (BASELINE configs[3]: a megabyte-scale transcript over several GPUs) ------------------
// Shard s covers blocks [s 2^k, (s + 1) 2^k) of the padded message; its proof says "from chaining value c_s these blocks lead to c_{s+1}"
// (the chained program: 32 public values).  c_0 = the standard IV, c_last = the digest.  Given the chaining values -- plain SHA-256
// compression on the host, one pass over the message -- the shards are INDEPENDENT: dealt over the devices like any other batch.
size_t zkhip_sha256_air_chained(uint32_t* program, size_t cap_words) {
    const std::vector<uint32_t>& p = sha::program_chained();
    if (program && cap_words >= p.size()) std::memcpy(program, p.data(), p.size() * 4);
    return p.size();
}
static int sharded_shape(size_t message_len, int log_blocks_per_shard, size_t* padded, size_t* n_shards, int* last_log_blocks) {
    if (log_blocks_per_shard < 0 || log_blocks_per_shard > 14) return fail(ZKHIP_ERR_INVALID, "sha256_sharded: 2^0 .. 2^14 blocks per shard (a shard is one chip proof of up to 2^20 rows)");
    *padded = ((message_len + 9 + 63) / 64) * 64;
    const size_t blocks = *padded / 64, per = (size_t)1 << log_blocks_per_shard;
    *n_shards = (blocks + per - 1) / per;
    const size_t rest = blocks - (*n_shards - 1) * per;           // blocks of the last shard: its trace holds the next power of two
    int lb = 0;
    while (((size_t)1 << lb) < rest) lb++;
    *last_log_blocks = lb;
    return *n_shards <= 4096 ? ZKHIP_OK : fail(ZKHIP_ERR_INVALID, "sha256_sharded: more than 4096 shards");
}
size_t zkhip_sha256_sharded_count(size_t message_len, int log_blocks_per_shard) {
    size_t padded, n;
    int lb;
    return sharded_shape(message_len, log_blocks_per_shard, &padded, &n, &lb) == ZKHIP_OK ? n : 0;
}
size_t zkhip_sha256_shard_proof_size(int log_blocks, const zkhip_params* prm) {
    const std::vector<uint32_t>& p = sha::program_chained();
    if (log_blocks < 0 || log_blocks > 14) return 0;
    return zkhip_proof_size_air(p.data(), p.size(), 6 + log_blocks, sha::WIDTH, prm, sha::N_PUBLIC_CHAINED);
}
static void chain_limbs(const uint32_t in[8], const uint32_t out[8], uint32_t pv[32]) {      // public values of a shard: final limbs, then initial limbs
    for (int i = 0; i < 8; i++) { pv[2 * i] = out[i] & 0xffffu; pv[2 * i + 1] = out[i] >> 16; pv[16 + 2 * i] = in[i] & 0xffffu; pv[16 + 2 * i + 1] = in[i] >> 16; }
}
// (uniform: the last shard at the full height too, its unused blocks inactive -- shards of ONE shape, which a join takes)
static int prove_sharded(const int* devices, int n_devices, const uint8_t* message, size_t message_len, int log_blocks_per_shard, const zkhip_params* prm,
                         int in_flight_per_device, uint8_t digest[32], uint32_t* chain, uint8_t* proofs, size_t proof_stride, size_t* proof_lens, bool uniform) {
    if ((message_len && !message) || !prm || !digest || !chain || !proofs || !proof_lens) return fail(ZKHIP_ERR_INVALID, "prove_sha256_sharded: null argument");
    size_t padded, n_shards;
    int last_lb;
    ZK_TRY(sharded_shape(message_len, log_blocks_per_shard, &padded, &n_shards, &last_lb));
    if (proof_stride < zkhip_sha256_shard_proof_size(log_blocks_per_shard, prm) || zkhip_sha256_shard_proof_size(log_blocks_per_shard, prm) == 0)
        return fail(ZKHIP_ERR_BUFFER, "prove_sha256_sharded: proof_stride is smaller than zkhip_sha256_shard_proof_size (or the proof shape is invalid)");
    std::vector<int> devs;
    ZK_TRY(resolve_devices(devices, n_devices, "prove_sha256_sharded", devs));
    std::vector<uint8_t> blocks(padded);
    zkhip_sha256_pad(message, message_len, blocks.data(), padded);
    // one pass of plain compression over the message gives the chaining value every shard starts from -- the only sequential part (about
    // 3 ms per MiB).  It runs on a thread of its own while the first shards are already being proven: shard s waits for chain[s + 1] only.
    const size_t per = (size_t)1 << log_blocks_per_shard, n_blocks = padded / 64;
    std::atomic<size_t> ready{0};                          // chain[0 .. ready) are final
    auto hash_all = [&] {
        uint32_t h[8];
        std::memcpy(h, sha::IV, 32);
        for (size_t k = 0; k < n_blocks; k++) {
            if (k % per == 0) { std::memcpy(chain + 8 * (k / per), h, 32); ready.store(k / per + 1, std::memory_order_release); }
            sha::compress(h, blocks.data() + 64 * k);
        }
        std::memcpy(chain + 8 * n_shards, h, 32);
        for (int i = 0; i < 8; i++) { digest[4 * i] = (uint8_t)(h[i] >> 24); digest[4 * i + 1] = (uint8_t)(h[i] >> 16); digest[4 * i + 2] = (uint8_t)(h[i] >> 8); digest[4 * i + 3] = (uint8_t)h[i]; }
        ready.store(n_shards + 1, std::memory_order_release);
    };
    std::thread hasher;
    try { hasher = std::thread(hash_all); } catch (...) { hash_all(); }      // no thread to be had: hash first, then prove (never throw across the C ABI)
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{hasher};
    const std::vector<uint32_t>& prog = sha::program_chained();
    std::vector<char> ran;
    for (size_t s = 0; s < n_shards; s++) proof_lens[s] = 0;
    return deal_jobs(devs.data(), (int)devs.size(), (int)n_shards, in_flight_per_device, [&](zkhip_ctx* ctx, int s) {
        while (ready.load(std::memory_order_acquire) < (size_t)s + 2) std::this_thread::yield();
        const size_t first = (size_t)s * per, active = (size_t)s + 1 == n_shards ? n_blocks - first : per;
        const int lb = (size_t)s + 1 == n_shards && !uniform ? last_lb : log_blocks_per_shard;
        void* trace;
        ZK_TRY(ctx_reserve(ctx, S_CHIP, ((size_t)sha::WIDTH << (6 + lb)) * 4, &trace));
        uint32_t out_pub[sha::N_PUBLIC], pv[sha::N_PUBLIC_CHAINED];
        ZK_TRY(zkhip_sha256_gen_trace_chained(ctx, chain + 8 * s, blocks.data() + 64 * first, active, (size_t)1 << lb, message_len, first, (uint32_t*)trace, sha::WIDTH, out_pub));
        chain_limbs(chain + 8 * s, chain + 8 * (s + 1), pv);
        for (int i = 0; i < 16; i++) if (out_pub[i] != pv[i]) return fail(ZKHIP_ERR_INTERNAL, "prove_sha256_sharded: a shard's trace does not end in the next chaining value");
        std::memcpy(pv + 2 * sha::N_DIGEST, out_pub + sha::N_DIGEST, sha::N_PAD * 4);            // this slice's padding values: the verifier derives the same from (length, shard)
        return zkhip_prove_shard_air(ctx, prog.data(), prog.size(), (const uint32_t*)trace, sha::WIDTH, 6 + lb, sha::WIDTH, pv, sha::N_PUBLIC_CHAINED, prm,
                                     proofs + (size_t)s * proof_stride, proof_stride, &proof_lens[s]);
    }, ran);
}
int zkhip_prove_sha256_sharded(const int* devices, int n_devices, const uint8_t* message, size_t message_len, int log_blocks_per_shard, const zkhip_params* prm,
                               int in_flight_per_device, uint8_t digest[32], uint32_t* chain, uint8_t* proofs, size_t proof_stride, size_t* proof_lens) {
    return prove_sharded(devices, n_devices, message, message_len, log_blocks_per_shard, prm, in_flight_per_device, digest, chain, proofs, proof_stride, proof_lens, false);
}
// checks a chain of shard proofs: chain[0] = the standard IV, chain[n] = the digest, shard s proves chain[s] -> chain[s + 1]; every shard but
// the last covers 2^log_blocks_per_shard blocks (the last proof's own header says how many rows it has).  *reason: the failing shard's check
int zkhip_verify_sha256_sharded(const uint8_t* proofs, size_t proof_stride, const size_t* proof_lens, size_t n_shards, const uint32_t* chain,
                                int log_blocks_per_shard, const uint8_t digest[32], uint64_t message_len, const zkhip_params* prm, size_t* bad_shard, int* reason) {
    if (!proofs || !proof_lens || !chain || !digest || !prm || n_shards < 1 || n_shards > 4096 || log_blocks_per_shard < 0 || log_blocks_per_shard > 14)
        return fail(ZKHIP_ERR_INVALID, "verify_sha256_sharded: bad arguments");
    const uint64_t per_shard = (uint64_t)1 << log_blocks_per_shard, total_blocks = (message_len + 8) / 64 + 1;
    if ((total_blocks + per_shard - 1) / per_shard != n_shards) {          // the statement's length fixes the number of shards
        if (bad_shard) *bad_shard = 0;
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_sha256_sharded: a message of this length has another number of shards");
    }
    if (bad_shard) *bad_shard = 0;
    if (reason) *reason = 0;
    auto reject = [&](size_t s, int why, const char* msg) { if (bad_shard) *bad_shard = s; if (reason) *reason = why; return fail(ZKHIP_ERR_VERIFY, msg); };
    if (std::memcmp(chain, sha::IV, 32) != 0) return reject(0, 1, "verify_sha256_sharded: the chain does not start from the SHA-256 initial value");
    for (int i = 0; i < 8; i++) {
        const uint32_t w = ((uint32_t)digest[4 * i] << 24) | ((uint32_t)digest[4 * i + 1] << 16) | ((uint32_t)digest[4 * i + 2] << 8) | digest[4 * i + 3];
        if (chain[8 * n_shards + i] != w) return reject(n_shards - 1, 1, "verify_sha256_sharded: the chain does not end in the digest");
    }
    const std::vector<uint32_t>& prog = sha::program_chained();
    // the shards are independent: a few host threads take them in turn (each verification spreads its queries over threads of its own);
    // the verdict is that of the LOWEST failing shard, whatever the order the threads finished in
    std::vector<int> rc(n_shards, ZKHIP_OK), why(n_shards, 0);
    std::vector<std::string> msg(n_shards);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t s = next.fetch_add(1);
            if (s >= n_shards) return;
            const uint8_t* pf = proofs + s * proof_stride;
            auto bad = [&](int w, const char* m) { rc[s] = ZKHIP_ERR_VERIFY; why[s] = w; msg[s] = m; };
            if (proof_lens[s] < 16 || proof_lens[s] > proof_stride) { bad(2, "verify_sha256_sharded: bad proof length"); continue; }
            uint32_t head[4];
            std::memcpy(head, pf, 16);
            const int log_n = (int)head[2];
            const bool last = s + 1 == n_shards;
            if (log_n < 6 || log_n > 20 || (!last && log_n != 6 + log_blocks_per_shard) || (last && log_n > 6 + log_blocks_per_shard)) {
                bad(3, "verify_sha256_sharded: a shard has the wrong height");
                continue;
            }
            uint32_t pv[sha::N_PUBLIC_CHAINED];
            chain_limbs(chain + 8 * s, chain + 8 * (s + 1), pv);
            const uint64_t first = (uint64_t)s * per_shard, active = last ? total_blocks - first : per_shard;
            if (active > ((uint64_t)1 << (log_n - 6))) { bad(3, "verify_sha256_sharded: a shard has the wrong height"); continue; }
            (void)sha::padding_publics(message_len, first, active, pv + 2 * sha::N_DIGEST);
            int w = 0;
            if (zkhip_verify_shard_air(prog.data(), prog.size(), pf, proof_lens[s], log_n, sha::WIDTH, pv, sha::N_PUBLIC_CHAINED, prm, &w) != ZKHIP_OK) { rc[s] = ZKHIP_ERR_VERIFY; why[s] = w; msg[s] = zkhip_last_error(); }
        }
    };
    const size_t nt = n_shards < 4 ? n_shards : 4;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; t++) { try { pool.emplace_back(work); } catch (...) { break; } }          // fewer threads, same verdict
    work();
    for (auto& t : pool) t.join();
    for (size_t s = 0; s < n_shards; s++)
        if (rc[s] != ZKHIP_OK) {
            if (bad_shard) *bad_shard = s;
            if (reason) *reason = why[s];
            return fail(ZKHIP_ERR_VERIFY, msg[s]);
        }
    return ZKHIP_OK;
}

// ---- the chain as ONE proof (core -> compress, sp1.rs:116, on a real statement): the shards -- all at the full height -- are verified in-circuit
// by the shard verifier machine in air mode (fri_chip.hip / shard_verifier.inl: the chained program's terms on the EVAL chip).  The outer
// proof's public values are the shards' (chaining values in and out, the slice's padding values): its verifier rebuilds them from the
// digest, the length and the chain, and checks that the chain starts from the initial value, links up and ends in the digest.
static int compressed_shape(size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer, size_t* n_shards) {
    if (!inner || !outer) return fail(ZKHIP_ERR_INVALID, "sha256_compressed: null argument");
    size_t padded;
    int last;
    return sharded_shape(message_len, log_blocks_per_shard, &padded, n_shards, &last);
}
static void compressed_publics(uint64_t message_len, int log_blocks_per_shard, size_t n_shards, const uint32_t* chain, std::vector<uint32_t>& pv) {
    const uint64_t per = (uint64_t)1 << log_blocks_per_shard, total = (message_len + 8) / 64 + 1;
    pv.assign(n_shards * sha::N_PUBLIC_CHAINED, 0u);
    for (size_t s = 0; s < n_shards; s++) {
        uint32_t* p = pv.data() + s * sha::N_PUBLIC_CHAINED;
        chain_limbs(chain + 8 * s, chain + 8 * (s + 1), p);
        const uint64_t first = (uint64_t)s * per;
        (void)sha::padding_publics(message_len, first, s + 1 == n_shards ? total - first : per, p + 2 * sha::N_DIGEST);
    }
}
int zkhip_sha256_compress_setup(zkhip_ctx* ctx, size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer,
                                zkhip_machine_key** key, uint32_t vk[8]) {
    size_t n;
    ZK_TRY(compressed_shape(message_len, log_blocks_per_shard, inner, outer, &n));
    const std::vector<uint32_t>& prog = sha::program_chained();
    return zkhip_shard_verifier_setup_air(ctx, prog.data(), prog.size(), 6 + log_blocks_per_shard, sha::WIDTH, inner->num_queries, inner->pow_bits, sha::N_PUBLIC_CHAINED, n, outer, key, vk);
}
int zkhip_sha256_compress_key_host(size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer, uint32_t vk[8]) {
    size_t n;
    ZK_TRY(compressed_shape(message_len, log_blocks_per_shard, inner, outer, &n));
    const std::vector<uint32_t>& prog = sha::program_chained();
    return zkhip_shard_verifier_key_host_air(prog.data(), prog.size(), 6 + log_blocks_per_shard, sha::WIDTH, inner->num_queries, inner->pow_bits, sha::N_PUBLIC_CHAINED, n, outer, vk);
}
size_t zkhip_sha256_compressed_proof_size(size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer) {
    size_t n;
    if (compressed_shape(message_len, log_blocks_per_shard, inner, outer, &n) != ZKHIP_OK) return 0;
    const std::vector<uint32_t>& prog = sha::program_chained();
    return zkhip_shard_verifier_proof_size_air(prog.data(), prog.size(), 6 + log_blocks_per_shard, sha::WIDTH, inner->num_queries, inner->pow_bits, sha::N_PUBLIC_CHAINED, n, outer);
}
int zkhip_prove_sha256_compressed(zkhip_ctx* ctx, const zkhip_machine_key* key, const int* devices, int n_devices, const uint8_t* message, size_t message_len,
                                  int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer, int in_flight_per_device, uint8_t digest[32],
                                  uint32_t* chain, uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!key || !chain || !digest || !proof || !len) return fail(ZKHIP_ERR_INVALID, "prove_sha256_compressed: null argument");
    size_t n;
    ZK_TRY(compressed_shape(message_len, log_blocks_per_shard, inner, outer, &n));
    const size_t stride = zkhip_sha256_shard_proof_size(log_blocks_per_shard, inner);
    if (stride == 0) return fail(ZKHIP_ERR_INVALID, "prove_sha256_compressed: bad inner proof shape");
    std::vector<uint8_t> shards;
    std::vector<size_t> lens(n, 0);
    std::vector<const uint8_t*> ptrs(n);
    try { shards.resize(n * stride); } catch (const std::bad_alloc&) { return fail(ZKHIP_ERR_NOMEM, "prove_sha256_compressed: no host memory for the shard proofs"); }
    ZK_TRY(prove_sharded(devices, n_devices, message, message_len, log_blocks_per_shard, inner, in_flight_per_device, digest, chain, shards.data(), stride, lens.data(), true));
    for (size_t s = 0; s < n; s++) ptrs[s] = shards.data() + s * stride;
    std::vector<uint32_t> pv;
    compressed_publics(message_len, log_blocks_per_shard, n, chain, pv);
    const std::vector<uint32_t>& prog = sha::program_chained();
    return zkhip_prove_shard_verifier_air(ctx, key, prog.data(), prog.size(), ptrs.data(), lens.data(), n, 6 + log_blocks_per_shard, sha::WIDTH, pv.data(), sha::N_PUBLIC_CHAINED,
                                          inner, outer, proof, cap, len);
}
int zkhip_verify_sha256_compressed(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const uint32_t* chain, int log_blocks_per_shard,
                                   const uint32_t vk[8], const zkhip_params* inner, const zkhip_params* outer, int* reason) {
    if (!proof || !digest || !chain || !vk) return fail(ZKHIP_ERR_INVALID, "verify_sha256_compressed: null argument");
    size_t n;
    ZK_TRY(compressed_shape((size_t)message_len, log_blocks_per_shard, inner, outer, &n));
    if (reason) *reason = 0;
    auto reject = [&](const char* msg) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, msg); };
    if (std::memcmp(chain, sha::IV, 32) != 0) return reject("verify_sha256_compressed: the chain does not start from the SHA-256 initial value");
    for (int i = 0; i < 8; i++) {
        const uint32_t w = ((uint32_t)digest[4 * i] << 24) | ((uint32_t)digest[4 * i + 1] << 16) | ((uint32_t)digest[4 * i + 2] << 8) | digest[4 * i + 3];
        if (chain[8 * n + i] != w) return reject("verify_sha256_compressed: the chain does not end in the digest");
    }
    std::vector<uint32_t> pv;                                   // (shard s's final value IS shard s + 1's initial one: both are chain[s + 1])
    compressed_publics(message_len, log_blocks_per_shard, n, chain, pv);
    const std::vector<uint32_t>& prog = sha::program_chained();
    return zkhip_verify_shard_recursive_air(prog.data(), prog.size(), proof, len, 6 + log_blocks_per_shard, sha::WIDTH, inner->num_queries, inner->pow_bits, pv.data(),
                                            sha::N_PUBLIC_CHAINED, n, vk, outer, reason);
}

}  // extern "C"
