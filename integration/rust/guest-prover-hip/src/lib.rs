//! MI355X (gfx950) shard prover backend for zktls: `zktls prove -p hip`.
//!
//! Layout mirrors `crates/guest-prover-sp1` (`sp1.rs`) and `crates/guest-prover-r0` (`prover.rs`):
//! a `ProverType` that travels through the process environment, a builder-style guest prover and a
//! `ZkProver` implementation that never lets a panic escape.
pub mod ffi;
mod prover;

pub use prover::{Backend, HipGuestProver, ProverType, Shard, ShardSource, SyntheticShards};
