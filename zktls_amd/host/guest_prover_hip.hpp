// guest_prover_hip.hpp -- C++ mirror of the reference's prover plug point, for the HIP backend.
//
// The reference is Rust and no Rust toolchain exists in the build image, so the host side
// above the C ABI is written in C++ with the same names, argument meaning and error
// behaviour as the reference's glue crate (the Rust twin a maintainer would add is in
// INTEGRATION.md):
//
//   trait ZkProver::prove(&mut self, input, guest_program) -> Result<(Vec<u8>, Vec<u8>)>
//       core/src/prelude.rs:12-18                       -> zktls::ZkProver::prove
//   enum ProverType { Mock, Local, Cuda, Network } + set_env  (SP1_PROVER)
//       crates/guest-prover-sp1/src/sp1.rs:10-30        -> zktls::ProverType, set_env()
//   SP1GuestProver::{new, mock, local, cuda, network}
//       sp1.rs:32-65                                    -> zktls::HipGuestProver::{mock, local, hip, network}
//   _panic_catched_prove: catch_unwind -> anyhow error  sp1.rs:80-100
//                                                       -> every exception becomes ProveResult.error
//   "proof of <= 4 bytes means no proof"                sp1.rs:128-130 -> same rule
//
//   RISC Zero twin (crates/guest-prover-r0/src/prover.rs): ProverType::set_env (:19-28: Mock -> RISC0_DEV_MODE=true,
//   Local/Cuda -> RISC0_PROVER=local, Network -> RISC0_PROVER=bonsai), Risc0GuestProver::{mock, local, cuda,
//   network} (:36-57), panic_catched_prover (:70-76), the <= 4-byte rule (:101-103)
//                                                       -> zktls::Risc0HipGuestProver (Backend::Risc0): same class,
//                                                          other environment variables, RISC-Zero-like proof shape
//
// What prove() does here: the zkVM executor that turns (input, ELF) into shard traces is
// third-party and out of scope.  Without a shard source Local / Hip mode therefore
// FAILS ("no shard source"): a caller applying the reference's rule "proof.len() > 4 means a real proof"
// (sp1.rs:128-130) must never be handed bytes that attest nothing about the guest.  The synthetic stand-in
// is an explicit opt-in, with_synthetic(plan): `shards` synthetic AIR-satisfying shards of 2^log_n x width
// generated on the device, seed and public values bound to zkhip_request_digest(CBOR input, ELF), every
// shard proven through libzkhip and verified (zkhip_verify_shard) like sp1.rs:120; the batch blob it returns
// carries the SYNTHETIC flag in its header.  Shards are dealt over the prover's device list exactly as
// zkhip_prove_shards_multi deals them (shard s -> devices[s mod n]).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace zktls {

enum class ProverType { Mock, Local, Hip, Network };
enum class Backend { Sp1, Risc0 };

// sp1.rs:20-29 / prover.rs:19-28: the mode travels to the SDK through the process environment
void set_env(ProverType mode);
void set_env_r0(ProverType mode);
const char* prover_type_name(ProverType mode);

struct GuestInput {
    std::vector<uint8_t> cbor;   // ciborium::into_writer(&input) bytes, sp1.rs:108-109
};

struct ProveResult {
    bool ok = false;
    std::string error;               // anyhow-style message when !ok
    std::vector<uint8_t> output;     // public values
    std::vector<uint8_t> proof;      // empty when the backend produced <= 4 bytes (mock)
    std::vector<uint8_t> vk;         // after setup(): the verifying key the proof was checked against (64 bytes, see SetupResult)
};

// `let (pk, vk) = client.setup(guest_program)` (sp1.rs:113).  The proving key stays inside the prover (device-resident preprocessed
// traces, their LDEs and Merkle tree: zkhip_machine_key, parked with the worker's context); vk is what a verifier needs:
// 8 LE words = the commitment to the machine's preprocessed tables, then 8 LE words = zkhip_request_digest("", guest_program).
struct SetupResult {
    bool ok = false;
    std::string error;
    std::vector<uint8_t> vk;
};

class ZkProver {
public:
    virtual ~ZkProver() = default;
    virtual ProveResult prove(const GuestInput& input, const std::vector<uint8_t>& guest_program) = 0;
};

struct ShardPlan {
    int log_n = 20;
    uint32_t width = 256;
    uint32_t shards = 1;
    int num_queries = 100;     // Backend::Risc0 uses RISC Zero's 50 queries / no PoW when these stay at the defaults
    int pow_bits = 16;
    uint32_t in_flight = 0;    // shards proven at the same time (own context + host thread each); 0: $ZKTLS_HIP_IN_FLIGHT or 4
};

// One chip of a synthetic MACHINE shard (SP1's real shard structure, sp1-core-machine, reference Cargo.lock:5822): 2^log_n rows x width columns
// [preprocessed | main] under the synthetic AIR, `pairs` LogUp pairs as an interaction table -- in-table (partner -1) or ACROSS two tables of one
// height that look each other up (partner = the other chip's index, mutual) --, `pre_width` leading columns committed once by setup (a multiple of 8)
struct ChipPlan {
    int log_n = 20;
    uint32_t width = 32;
    uint32_t pairs = 1;
    int partner = -1;
    uint32_t pre_width = 0;
};
// An execution of `shards` such shards, each ONE version-11 proof of the keyed machine (zkhip_prove_machine_keyed) against ONE key: the
// preprocessed columns depend on the guest program alone (sp1.rs:113: setup(guest_program)), the main columns on (input, program, shard).
struct MachinePlan {
    std::vector<ChipPlan> chips;   // tallest first, at most 16
    uint32_t shards = 1;
    int num_queries = 100;
    int pow_bits = 16;
    uint32_t in_flight = 0;        // as ShardPlan::in_flight
    // the six chips bench.py's `multichip` section proves: 2^20 x 96 and 2^20 x 32 looking each other up, 2^19 x 64, 2^18 x 128, 2^16 x 256 with 32
    // preprocessed columns, 2^14 x 40; LogUp pairs 3, 3, 2, 4, 8, 1 (zktls_amd.device.SP1_SHAPED_SPEC)
    static MachinePlan sp1_shaped(uint32_t shards = 1);
};

class HipGuestProver : public ZkProver {
public:
    explicit HipGuestProver(int device = 0, Backend backend = Backend::Sp1) : devices_{device}, backend_(backend) {}
    // every shard of a request goes to devices[s mod n]; an empty list is refused at prove time
    HipGuestProver& with_devices(const std::vector<int>& devices) { devices_ = devices; return *this; }
    HipGuestProver& mock() { mode_ = ProverType::Mock; return *this; }
    HipGuestProver& local() { mode_ = ProverType::Local; return *this; }
    HipGuestProver& hip() { mode_ = ProverType::Hip; return *this; }
    HipGuestProver& network() { mode_ = ProverType::Network; return *this; }
    // explicit opt-in to the synthetic shard plan (no zkVM executor wired): see the header comment
    HipGuestProver& with_synthetic(const ShardPlan& p) { plan_ = p; synthetic_ = true; return *this; }
    // ... or to synthetic shards in SP1's shard STRUCTURE: several chips of mixed heights, lookups inside and across tables, preprocessed columns --
    // every shard one keyed-machine proof, checked against the key like sp1.rs:120 (blob flags SYNTHETIC | MACHINE; the result's vk = the key's
    // root + the program's digest, what setup() returns).  with_compress(): the shard proofs are verified in-circuit in machine mode
    // (zkhip_prove_machine_verifier: lookups, mixed heights, the preprocessed openings against the key) -- ONE proof while the shards fit one join
    // (64, or a Poseidon2 chip of 2^22 rows), else equal joins and ONE proof above them (flag TREE).  verify_machine_blob checks either blob on the host.
    HipGuestProver& with_synthetic_machine(const MachinePlan& p) { mplan_ = p; machine_ = true; synthetic_ = true; return *this; }
    // the input-commitment guest (see the header comment); num_queries / pow_bits of the proof come from `p`
    HipGuestProver& with_input_commitment(const ShardPlan& p = ShardPlan{}) { plan_ = p; commitment_ = true; return *this; }
    // the COMPRESS stage behind the same call (sp1.rs:116: core -> compress; prover.rs:90: lift -> join): after the shards are proven, ONE
    // proof verifies them all in-circuit (zkhip_prove_shard_verifier) and replaces them in the blob (flag COMPRESSED: entry 0 = the joined
    // proof, entry 1 = 8 LE words of the shape's key + the shard count).  SP1 backend, synthetic shards, width a multiple of 8.  More shards
    // than one join holds (136 of the headline shape) take several joins of ONE shape (compress_join_size) and ONE more proof above them (the tree; blob flag TREE).  verify_compressed_blob checks
    // such a blob on the host from (plan, input, ELF, key): the shard proofs are gone.
    // With with_input_commitment() and an input beyond one chip proof (1 MiB): the chain of SHA-256 shard proofs becomes ONE proof the same way
    // (zkhip_prove_sha256_compressed; blob flags INPUT_SHA256 | CHAINED | COMPRESSED) -- verify_commitment_blob checks it from (output, blob) alone.
    HipGuestProver& with_compress() { compress_ = true; return *this; }
    bool synthetic() const { return synthetic_; }
    ProverType mode() const { return mode_; }
    Backend backend() const { return backend_; }
    ProveResult prove(const GuestInput& input, const std::vector<uint8_t>& guest_program) override;
    // sp1.rs:113.  For the input-commitment guest on the SP1 backend: commits the SHA-256 machine's range table on the device (once per
    // parked context) and switches prove() to the keyed machine (chip + table, proof version 11, blob flag KEYED), verified against vk
    // like sp1.rs:120.  Mock mode returns the program digest with a zero commitment.  Other guests have no preprocessed tables: an error.
    SetupResult setup(const std::vector<uint8_t>& guest_program);

private:
    ProveResult prove_inner(const GuestInput& input, const std::vector<uint8_t>& guest_program);
    ProverType mode_ = ProverType::Mock;   // #[default] Mock, sp1.rs:12-13
    std::vector<int> devices_;
    Backend backend_ = Backend::Sp1;
    ShardPlan plan_;
    MachinePlan mplan_;
    bool machine_ = false;
    bool synthetic_ = false;
    bool commitment_ = false;
    bool compress_ = false;
    std::vector<uint8_t> vk_;              // non-empty after setup()
};

// prover.rs:30-57: `Risc0GuestProver::default().local()` etc.; segments are proven in RISC Zero's shape
// (blowup 4, fold 16, 256 final coefficients -- or fewer for tiny segments --, Poseidon2 width 24)
class Risc0HipGuestProver : public HipGuestProver {
public:
    explicit Risc0HipGuestProver(int device = 0) : HipGuestProver(device, Backend::Risc0) {}
};

// workers park their context and trace buffer for the next request; this frees them
void release_cached();

// 8 canonical BabyBear words binding (input, ELF): the public values of every shard
std::vector<uint32_t> request_digest(const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf);

// batch proof container: "ZKTB", version 2, flags (bit 0: SYNTHETIC shards, attests nothing about a guest; bit 1: INPUT_SHA256), shard count,
// then per shard (u32 length, bytes)
constexpr uint32_t BATCH_FLAG_SYNTHETIC = 1u;
constexpr uint32_t BATCH_FLAG_INPUT_SHA256 = 2u;     // one proof of the SHA-256 chip over the request's input bytes; the LAST entry is the input's length (8 LE bytes): with the padding constrained in-circuit the statement is "output = SHA-256 of a message of this length"
constexpr uint32_t BATCH_FLAG_CHAINED = 8u;          // with INPUT_SHA256: an input beyond one chip proof (1 MiB): entry 0 = the chaining values ((n + 1) x 8 LE words), entries 1..n = the shard proofs of zkhip_prove_sha256_sharded (2^14 blocks per shard)
constexpr uint32_t BATCH_FLAG_COMPRESSED = 16u;      // with SYNTHETIC: the shard proofs were joined into ONE proof (entry 0; k proofs when the shards do not fit one join); last entry = the shape's key (8 LE words) + the shard count
//                                                      with INPUT_SHA256 | CHAINED (with_input_commitment().with_compress(), inputs beyond 1 MiB): entry 0 = the chaining values, entry 1 = the ONE
//                                                      proof that verifies the shard proofs in-circuit (zkhip_prove_sha256_compressed), entry 2 = its key (8 LE words; a verifier derives it itself), then the length
// shard proofs per join for an execution of `shards` shards of `plan`'s shape: one join while they fit (zkhip_shard_verifier_max_proofs: the
// Poseidon2 chip's 2^22 rows -- 136 proofs of the headline shape); beyond, ceil(shards / max) joins of equal size J = ceil(shards / joins) --
// the last one repeats the execution's last shard proof to fill its J places, so that every join has the same shape, hence the same key
uint32_t compress_join_size(const ShardPlan& plan);
// ... or fewer per join (0: the most): an execution then takes several joins, and ONE more proof verifies the joins (machine mode of the shard
// verifier machine, csrc/machine_verifier.inl): the blob carries that proof alone and the flag TREE
void set_compress_join_size(uint32_t shard_proofs_per_join);
constexpr uint32_t BATCH_FLAG_TREE = 32u;            // with COMPRESSED: several joins were needed and were joined again: entry 0 = the ONE proof above them
constexpr uint32_t BATCH_FLAG_MACHINE = 64u;         // with SYNTHETIC: every shard is a keyed-MACHINE proof (MachinePlan); with COMPRESSED the joins are machine-mode proofs and the last entry = the join's key (8 LE words) + the shard count
constexpr uint32_t BATCH_FLAG_KEYED = 4u;            // with INPUT_SHA256: the proof is the keyed SHA-256 MACHINE's (chip + range table), checked against a vk
// a consumer's check of an input-commitment blob on the CPU: the blob's proof(s) against the claimed output (SHA-256 of the input).
// The caller says what it EXPECTS, the blob's own flags only have to agree: a 64-byte `vk` (from setup) means "a KEYED proof under this
// key" -- a plain or CHAINED blob is then refused (*reason = 2) instead of being accepted with the key ignored; without a vk a KEYED blob
// is refused.  `backend` picks the proof shape the prover used (Backend::Sp1: blowup 2, `num_queries`, `pow_bits`; Backend::Risc0: its
// segment shape, fold 16).  Bytes 32..63 of a vk are the guest program's digest: the PROVER refuses a program that differs from the one
// setup() saw, the proof itself does not bind it.  -> 0 or a negative value; *reason as the zkhip verifiers
// the input length a commitment blob states (its last entry, 8 LE bytes): the proofs attest "output = SHA-256 of a message of exactly this length"
bool commitment_blob_length(const std::vector<uint8_t>& blob, uint64_t* message_len);
int verify_commitment_blob(const std::vector<uint8_t>& blob, const std::vector<uint8_t>& output, const std::vector<uint8_t>& vk,
                           int num_queries, int pow_bits, int* reason = nullptr, Backend backend = Backend::Sp1);
// a COMPRESSED blob against the request it was made for: the public values of shard s are request_digest(cbor, elf) | s, the key must be the
// caller's (compress_key: the key of the plan's shape, computed on the device once -- it depends on no proof).  0, or -1 (malformed / another
// key) / -2 (the joined proof is rejected; *reason = the failing check)
int verify_compressed_blob(const std::vector<uint8_t>& blob, const ShardPlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf,
                           const uint32_t key[8], int* reason = nullptr);
// core -> compress as a step of its own (sp1-cuda's prove_core / compress pair behind sp1.rs:116): the SYNTHETIC batch blob prove() returned for
// (plan, input, ELF) -> the COMPRESSED (| TREE) blob with_compress() would have returned; verify_compressed_blob checks it the same way
ProveResult compress_blob(int device, const ShardPlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf, const std::vector<uint8_t>& blob);
// the key of the shard-verifier machine for `plan` (zkhip_shard_verifier_setup on `device`)
bool compress_key(int device, const ShardPlan& plan, uint32_t key[8], std::string* error = nullptr);
// the same key computed on the host's cores (no device, no context): a verifier that owns no GPU checks a compressed blob with this and verify_compressed_blob
bool compress_key_host(const ShardPlan& plan, uint32_t key[8], std::string* error = nullptr);
// A MACHINE blob against the request it was made for, on the host: vk = the 64 bytes setup() / prove() returned (the machine key's root + the program's
// digest, which must be request_digest("", elf)).  Plain: every shard proof under (the plan's machine, digest | s, the key).  COMPRESSED: the join's key
// -- and with TREE the top's -- is DERIVED here from (the plan, the key's root, the shard count): no byte of a shard proof is needed.
// 0, or -1 (malformed / another plan / another program) / -2 (a proof is rejected; *reason = the failing check)
int verify_machine_blob(const std::vector<uint8_t>& blob, const MachinePlan& plan, const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf,
                        const std::vector<uint8_t>& vk, int* reason = nullptr);
// shard proofs per machine-mode join for `plan` (the most one join takes, bounded by set_compress_join_size; equal joins beyond)
uint32_t machine_join_size(const MachinePlan& plan);
std::vector<uint8_t> pack_shard_proofs(const std::vector<std::vector<uint8_t>>& proofs, uint32_t flags);
bool unpack_shard_proofs(const std::vector<uint8_t>& blob, std::vector<std::vector<uint8_t>>* proofs, uint32_t* flags = nullptr);

}  // namespace zktls
