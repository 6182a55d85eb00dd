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
// What prove() does here: the zkVM executor that turns (input, ELF) into shard traces is
// third-party and out of scope, so this mirror derives the shard
// list deterministically from the request -- `shards` synthetic shards of 2^log_n x width
// whose seed and public values are bound to a digest of the CBOR input and the ELF -- and
// proves every shard through libzkhip (zkhip_gen_trace + zkhip_prove_shard), verifying each
// proof (zkhip_verify_shard) like sp1.rs:120.  Swap `plan_shards` for the real executor's
// output and the rest is unchanged.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace zktls {

enum class ProverType { Mock, Local, Hip, Network };

// sp1.rs:20-29: the mode travels to the SDK through the process environment
void set_env(ProverType mode);
const char* prover_type_name(ProverType mode);

struct GuestInput {
    std::vector<uint8_t> cbor;   // ciborium::into_writer(&input) bytes, sp1.rs:108-109
};

struct ProveResult {
    bool ok = false;
    std::string error;               // anyhow-style message when !ok
    std::vector<uint8_t> output;     // public values
    std::vector<uint8_t> proof;      // empty when the backend produced <= 4 bytes (mock)
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
    int num_queries = 100;
    int pow_bits = 16;
};

class HipGuestProver : public ZkProver {
public:
    explicit HipGuestProver(int device = 0) : device_(device) {}
    HipGuestProver& mock() { mode_ = ProverType::Mock; return *this; }
    HipGuestProver& local() { mode_ = ProverType::Local; return *this; }
    HipGuestProver& hip() { mode_ = ProverType::Hip; return *this; }
    HipGuestProver& network() { mode_ = ProverType::Network; return *this; }
    HipGuestProver& with_plan(const ShardPlan& p) { plan_ = p; return *this; }
    ProverType mode() const { return mode_; }
    ProveResult prove(const GuestInput& input, const std::vector<uint8_t>& guest_program) override;

private:
    ProveResult prove_inner(const GuestInput& input, const std::vector<uint8_t>& guest_program);
    ProverType mode_ = ProverType::Mock;   // #[default] Mock, sp1.rs:12-13
    int device_ = 0;
    ShardPlan plan_;
};

// 8 canonical BabyBear words binding (input, ELF): the public values of every shard
std::vector<uint32_t> request_digest(const std::vector<uint8_t>& cbor, const std::vector<uint8_t>& elf);

// batch proof container: "ZKTB", version, shard count, then per shard (u32 length, bytes)
std::vector<uint8_t> pack_shard_proofs(const std::vector<std::vector<uint8_t>>& proofs);
bool unpack_shard_proofs(const std::vector<uint8_t>& blob, std::vector<std::vector<uint8_t>>* proofs);

}  // namespace zktls
