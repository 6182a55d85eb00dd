//! `HipGuestProver`: the `ZkProver` of the HIP backend.
//!
//! Shape of the reference glue it stands next to:
//!   crates/guest-prover-sp1/src/sp1.rs  -- ProverType::set_env (:20-29), builder (:32-65), ZkProver impl (:66-78),
//!                                          catch_unwind (:85), cbor input (:108-109), verify (:120), <= 4-byte rule (:128-130)
//!   crates/guest-prover-r0/src/prover.rs -- the same for RISC Zero (:19-28, :60-67, :74, :81-86, :101-103)
use std::{future::Future, panic};

use anyhow::{anyhow, Result};
use zktls_core::ZkProver;
use zktls_program_core::GuestInput;

use crate::ffi::{self, check, Context, ZkhipParams};

#[derive(Default, Clone, Copy, PartialEq, Eq, Debug)]
pub enum ProverType {
    #[default]
    Mock,
    Local,
    Hip,
    Network,
}

/// Which SDK's conventions (environment variables, proof shape) the backend follows.
#[derive(Default, Clone, Copy, PartialEq, Eq, Debug)]
pub enum Backend {
    #[default]
    Sp1,
    Risc0,
}

impl ProverType {
    pub fn set_env(&self, backend: Backend) {
        match (backend, self) {
            (Backend::Sp1, ProverType::Mock) => std::env::set_var("SP1_PROVER", "mock"),
            (Backend::Sp1, ProverType::Local) => std::env::set_var("SP1_PROVER", "local"),
            (Backend::Sp1, ProverType::Hip) => std::env::set_var("SP1_PROVER", "hip"),
            (Backend::Sp1, ProverType::Network) => std::env::set_var("SP1_PROVER", "network"),
            (Backend::Risc0, ProverType::Mock) => std::env::set_var("RISC0_DEV_MODE", "true"),
            (Backend::Risc0, ProverType::Local) | (Backend::Risc0, ProverType::Hip) => std::env::set_var("RISC0_PROVER", "local"),
            (Backend::Risc0, ProverType::Network) => std::env::set_var("RISC0_PROVER", "bonsai"),
        }
    }
}

/// One shard / segment trace as the zkVM executor hands it over.
pub struct Shard {
    pub log_n: i32,
    pub width: u32,
    /// row-major (`column_major == false`, SP1 `RowMajorMatrix`) or `width` contiguous columns (RISC Zero Hal)
    pub column_major: bool,
    /// canonical BabyBear words, `width << log_n` of them
    pub values: Vec<u32>,
    pub public_values: Vec<u32>,
}

/// The executor side (sp1-core-executor / risc0 executor): turns (cbor input, ELF) into shard traces and
/// the guest's public output.  Out of scope of the HIP library; plug the real executor in here.
pub trait ShardSource: Send {
    fn shards(&mut self, input_cbor: &[u8], elf: &[u8]) -> Result<(Vec<u8>, Vec<Shard>)>;
}

/// Stand-in used until the executor is wired: `count` synthetic AIR-satisfying shards generated on the device
/// (`zkhip_gen_trace`), seed and public values bound to `zkhip_request_digest(cbor, elf)`.  EXPLICIT OPT-IN only
/// (`with_synthetic`): such a blob attests nothing about the guest, its header carries `BATCH_FLAG_SYNTHETIC`.
pub struct SyntheticShards {
    pub log_n: i32,
    pub width: u32,
    pub count: u32,
}

pub struct HipGuestProver {
    mode: ProverType,
    backend: Backend,
    device: i32,
    source: Option<Box<dyn ShardSource>>,
    synthetic: Option<SyntheticShards>,
    commitment: bool,
    vk: Option<Vec<u8>>,               // Some after setup()
    keyed: Option<KeyedContext>,       // the proving key of setup() (None in mock mode)
}

/// batch blob header: "ZKTB", version 2, flags, shard count; then (u32 length, bytes) per shard
pub const BATCH_FLAG_SYNTHETIC: u32 = 1;
/// one proof of the SHA-256 chip over the request's input bytes (`with_input_commitment`)
pub const BATCH_FLAG_INPUT_SHA256: u32 = 2;
/// with INPUT_SHA256: the proof is the keyed SHA-256 machine's (chip + range table), to be checked against the vk of `setup`
pub const BATCH_FLAG_KEYED: u32 = 4;
/// with INPUT_SHA256: an input beyond one chip proof -- entry 0 holds the chaining values, entries 1..n the shard proofs (C++ mirror: zkhip_prove_sha256_sharded)
pub const BATCH_FLAG_CHAINED: u32 = 8;

/// the proving key of `setup`: a context and the machine key made on it (destroyed together, key first)
pub struct KeyedContext {
    ctx: Context,
    key: *mut ffi::ZkhipMachineKey,
}
impl Drop for KeyedContext {
    fn drop(&mut self) {
        unsafe { ffi::zkhip_machine_key_destroy(self.key) };
    }
}
unsafe impl Send for KeyedContext {}

impl HipGuestProver {
    pub fn new(device: i32) -> Self {
        Self { mode: ProverType::default(), backend: Backend::default(), device, source: None, synthetic: None, commitment: false, vk: None, keyed: None }
    }
    pub fn mock(mut self) -> Self { self.mode = ProverType::Mock; self }
    pub fn local(mut self) -> Self { self.mode = ProverType::Local; self }
    pub fn hip(mut self) -> Self { self.mode = ProverType::Hip; self }
    pub fn network(mut self) -> Self { self.mode = ProverType::Network; self }
    pub fn risc0(mut self) -> Self { self.backend = Backend::Risc0; self }
    pub fn with_source(mut self, source: Box<dyn ShardSource>) -> Self { self.source = Some(source); self }
    /// opt into proving synthetic shards when no executor is wired (see `SyntheticShards`)
    pub fn with_synthetic(mut self, plan: SyntheticShards) -> Self { self.synthetic = Some(plan); self }
    /// the input-commitment guest: prove "I know the request's input and its SHA-256 is `output`" through the SHA-256
    /// compression chip of libzkhip -- a real statement that needs no executor (not the zkTLS verifier guest)
    pub fn with_input_commitment(mut self) -> Self { self.commitment = true; self }

    fn params(&self, log_n: i32) -> ZkhipParams {
        match self.backend {
            Backend::Sp1 => ZkhipParams::SP1_CORE,
            Backend::Risc0 => {
                // the final polynomial shrinks for segments too small for 256 coefficients
                let mut lf = 8;
                while lf > log_n || (log_n - lf) % 4 != 0 { lf -= 1; }
                ZkhipParams { log_final: lf, ..ZkhipParams::RISC0 }
            }
        }
    }
}

impl ZkProver for HipGuestProver {
    fn prove(&mut self, input: GuestInput, guest_program: &[u8]) -> impl Future<Output = Result<(Vec<u8>, Vec<u8>)>> + Send {
        self.mode.set_env(self.backend);
        let elf = guest_program.to_vec();                       // copied once, like sp1.rs:74
        let result = (|| -> Result<(Vec<u8>, Vec<u8>)> {
            let mut cbor = Vec::new();
            ciborium::into_writer(&input, &mut cbor)?;          // sp1.rs:108-109 / prover.rs:81-82
            log::info!("input_len: {}", cbor.len());
            let this = panic::AssertUnwindSafe(&mut *self);
            panic::catch_unwind(move || { let this = this; this.0.prove_blocking(&cbor, &elf) })
                .map_err(|e| anyhow!("{:?}", e))?               // sp1.rs:85: nothing unwinds past the glue
        })();
        async move { result }
    }
}

impl HipGuestProver {
    /// `let (pk, vk) = client.setup(guest_program)` (sp1.rs:113) for the input-commitment guest: commits the SHA-256 machine's range table
    /// on the device; the proving key stays in the prover, the returned 64 bytes are the verifying key (8 LE words commitment, 8 LE words
    /// request digest of the program), as the C++ mirror's `HipGuestProver::setup`.  After it, `prove` produces the keyed machine's proof
    /// (blob flags INPUT_SHA256 | KEYED) and checks it against this vk.
    pub fn setup(&mut self, guest_program: &[u8]) -> Result<Vec<u8>> {
        anyhow::ensure!(!guest_program.is_empty(), "guest program is empty");
        anyhow::ensure!(self.commitment, "setup: only the input-commitment guest has preprocessed tables (with_input_commitment())");
        let mut pd = [0u32; 8];
        check(unsafe { ffi::zkhip_request_digest(std::ptr::null(), 0, guest_program.as_ptr(), guest_program.len(), pd.as_mut_ptr()) }, "zkhip_request_digest")?;
        let mut vk = vec![0u8; 32];
        if !matches!(self.mode, ProverType::Mock) {
            anyhow::ensure!(!matches!(self.mode, ProverType::Network), "network proving is not provided by the HIP backend");
            let ctx = Context::new(self.device)?;
            let prm = self.params(16);
            let (mut key, mut root) = (std::ptr::null_mut(), [0u32; 8]);
            check(unsafe { ffi::zkhip_sha256_setup(ctx.raw(), &prm, &mut key, root.as_mut_ptr()) }, "zkhip_sha256_setup")?;
            vk = root.iter().flat_map(|w| w.to_le_bytes()).collect();
            self.keyed = Some(KeyedContext { ctx, key });                  // pk: lives as long as the prover
        }
        vk.extend(pd.iter().flat_map(|w| w.to_le_bytes()));
        self.vk = Some(vk.clone());
        Ok(vk)
    }

    fn prove_blocking(&mut self, cbor: &[u8], elf: &[u8]) -> Result<(Vec<u8>, Vec<u8>)> {
        anyhow::ensure!(!elf.is_empty(), "guest program is empty");
        // the request digest: 8 canonical words = the leading public values of every shard and, without an executor, the
        // public output -- the same function the C++ mirror calls (zktls_amd/host/guest_prover_hip.cpp)
        let mut digest = [0u32; 8];
        check(unsafe { ffi::zkhip_request_digest(cbor.as_ptr(), cbor.len(), elf.as_ptr(), elf.len(), digest.as_mut_ptr()) }, "zkhip_request_digest")?;
        let digest_bytes: Vec<u8> = digest.iter().flat_map(|w| w.to_le_bytes()).collect();
        match self.mode {
            // executes nothing: public output = the digest, proof = a <= 4-byte placeholder, i.e. "no proof" (sp1.rs:128-130)
            ProverType::Mock if self.commitment => {
                let mut d = [0u8; 32];
                unsafe { ffi::zkhip_sha256_digest(cbor.as_ptr(), cbor.len(), d.as_mut_ptr()) };
                return Ok((d.to_vec(), Vec::new()));
            }
            ProverType::Mock => return Ok((digest_bytes, Vec::new())),
            ProverType::Network => return Err(anyhow!("network proving is not provided by the HIP backend")),
            ProverType::Local | ProverType::Hip => {}
        }
        // A caller applying the reference's rule "proof.len() > 4 means a real proof" must never receive bytes that attest
        // nothing about the guest: without an executor the synthetic plan is an explicit opt-in, otherwise this is an error.
        if self.source.is_none() && self.synthetic.is_none() && !self.commitment {
            return Err(anyhow!("no shard source: wire the zkVM executor with with_source(), or opt into synthetic shards with with_synthetic() / the input-commitment guest with with_input_commitment()"));
        }
        if unsafe { ffi::zkhip_device_count() } <= 0 {
            return Err(anyhow!("no gfx950 device: libzkhip has no CPU fallback"));
        }
        let ctx = Context::new(self.device)?;
        let start = std::time::Instant::now();
        let mut proofs: Vec<Vec<u8>> = Vec::new();
        let (output, flags) = if self.commitment && self.vk.is_some() {
            // after setup(): the keyed SHA-256 machine (chip + preprocessed range table), verified against the vk (sp1.rs:120)
            let vk = self.vk.as_ref().unwrap();
            let mut pd = [0u32; 8];
            check(unsafe { ffi::zkhip_request_digest(std::ptr::null(), 0, elf.as_ptr(), elf.len(), pd.as_mut_ptr()) }, "zkhip_request_digest")?;
            let pd_bytes: Vec<u8> = pd.iter().flat_map(|w| w.to_le_bytes()).collect();
            anyhow::ensure!(vk[32..] == pd_bytes[..], "prove: the guest program is not the one setup() was called with");
            let keyed = self.keyed.as_ref().ok_or_else(|| anyhow!("prove: setup() ran in mock mode"))?;
            let prm = self.params(16);
            let cap = unsafe { ffi::zkhip_sha256_machine_proof_size(cbor.len(), &prm) };
            anyhow::ensure!(cap > 0, "input too long for the SHA-256 machine");
            let (mut proof, mut len, mut d) = (vec![0u8; cap], 0usize, [0u8; 32]);
            check(unsafe { ffi::zkhip_prove_sha256_machine(keyed.ctx.raw(), keyed.key, cbor.as_ptr(), cbor.len(), &prm, d.as_mut_ptr(), proof.as_mut_ptr(), cap, &mut len) }, "zkhip_prove_sha256_machine")?;
            proof.truncate(len);
            let mut reason = 0;
            check(unsafe { ffi::zkhip_verify_sha256_machine(proof.as_ptr(), proof.len(), d.as_ptr(), cbor.len() as u64, vk.as_ptr() as *const u32, &prm, &mut reason) }, "zkhip_verify_sha256_machine")?;
            proofs.push(proof);
            proofs.push((cbor.len() as u64).to_le_bytes().to_vec());       // the statement's other half: the input's length (the padding is constrained in-circuit)
            (d.to_vec(), BATCH_FLAG_INPUT_SHA256 | BATCH_FLAG_KEYED)
        } else if self.commitment && cbor.len() > (1usize << 20) - 9 && matches!(self.backend, Backend::Sp1) {
            // a large transcript: SHA-256 as a CHAIN of shard proofs over the device list (as the C++ mirror): entry 0 of the blob
            // holds the chaining values, entries 1..n the shards; blob flags INPUT_SHA256 | CHAINED
            let k = 14;
            let prm = self.params(20);
            let n = unsafe { ffi::zkhip_sha256_sharded_count(cbor.len(), k) };
            let stride = unsafe { ffi::zkhip_sha256_shard_proof_size(k, &prm) };
            anyhow::ensure!(n > 0 && stride > 0, "input commitment: bad shard shape");
            let (mut buf, mut lens, mut chain, mut d) = (vec![0u8; n * stride], vec![0usize; n], vec![0u32; (n + 1) * 8], [0u8; 32]);
            let devices = [self.device];
            check(unsafe { ffi::zkhip_prove_sha256_sharded(devices.as_ptr(), 1, cbor.as_ptr(), cbor.len(), k, &prm, 2, d.as_mut_ptr(), chain.as_mut_ptr(), buf.as_mut_ptr(), stride, lens.as_mut_ptr()) }, "zkhip_prove_sha256_sharded")?;
            let (mut bad, mut reason) = (0usize, 0);
            check(unsafe { ffi::zkhip_verify_sha256_sharded(buf.as_ptr(), stride, lens.as_ptr(), n, chain.as_ptr(), k, d.as_ptr(), cbor.len() as u64, &prm, &mut bad, &mut reason) }, "zkhip_verify_sha256_sharded")?;   // sp1.rs:120
            proofs.push(chain.iter().flat_map(|w| w.to_le_bytes()).collect());
            for s in 0..n {
                proofs.push(buf[s * stride..s * stride + lens[s]].to_vec());
            }
            proofs.push((cbor.len() as u64).to_le_bytes().to_vec());
            (d.to_vec(), BATCH_FLAG_INPUT_SHA256 | BATCH_FLAG_CHAINED)
        } else if self.commitment {
            // 64 rows per 64-byte block, block count (padding included) rounded up to a power of two
            let blocks = (cbor.len() + 9 + 63) / 64;
            let log_n = 6 + (blocks.next_power_of_two().trailing_zeros() as i32);
            let prm = self.params(log_n);
            let cap = unsafe { ffi::zkhip_sha256_proof_size(cbor.len(), &prm) };
            anyhow::ensure!(cap > 0, "input too long for the SHA-256 chip");
            let (mut proof, mut len, mut d) = (vec![0u8; cap], 0usize, [0u8; 32]);
            check(unsafe { ffi::zkhip_prove_sha256(ctx.raw(), cbor.as_ptr(), cbor.len(), &prm, d.as_mut_ptr(), proof.as_mut_ptr(), cap, &mut len) }, "zkhip_prove_sha256")?;
            proof.truncate(len);
            let mut reason = 0;
            check(unsafe { ffi::zkhip_verify_sha256(proof.as_ptr(), proof.len(), d.as_ptr(), cbor.len() as u64, &prm, &mut reason) }, "zkhip_verify_sha256")?;   // sp1.rs:120
            proofs.push(proof);
            proofs.push((cbor.len() as u64).to_le_bytes().to_vec());
            (d.to_vec(), BATCH_FLAG_INPUT_SHA256)
        } else if let Some(src) = self.source.as_mut() {
            let (output, shards) = src.shards(cbor, elf)?;
            for shard in &shards {
                let prm = self.params(shard.log_n);
                let buf = ctx.alloc(shard.values.len())?;
                buf.upload_canonical(&shard.values)?;
                let mut pv = digest.to_vec();
                pv.extend_from_slice(&shard.public_values);
                proofs.push(prove_one(&ctx, buf.ptr, shard.log_n, shard.width, shard.column_major, &pv, &prm)?);
            }
            (output, 0u32)
        } else {
            let plan = self.synthetic.as_ref().unwrap();
            let prm = self.params(plan.log_n);
            let words = (plan.width as usize) << plan.log_n;
            let buf = ctx.alloc(words)?;
            let seed = digest[..4].iter().fold(0u64, |h, w| (h << 16) ^ (*w as u64));      // as the C++ mirror
            for s in 0..plan.count {
                check(unsafe { ffi::zkhip_gen_trace(ctx.raw(), seed, s as u64, plan.log_n, plan.width, buf.ptr, plan.width as usize) }, "zkhip_gen_trace")?;
                let mut pv = digest.to_vec();
                pv.push(s);
                proofs.push(prove_one(&ctx, buf.ptr, plan.log_n, plan.width, false, &pv, &prm)?);
            }
            (digest_bytes, BATCH_FLAG_SYNTHETIC)
        };
        let mut blob = Vec::new();
        blob.extend_from_slice(&0x42544B5Au32.to_le_bytes());               // "ZKTB"
        blob.extend_from_slice(&2u32.to_le_bytes());
        blob.extend_from_slice(&flags.to_le_bytes());
        blob.extend_from_slice(&(proofs.len() as u32).to_le_bytes());
        for p in &proofs {
            blob.extend_from_slice(&(p.len() as u32).to_le_bytes());
            blob.extend_from_slice(p);
        }
        log::info!("Proving took: {:?}", start.elapsed());
        if blob.len() <= 4 {
            blob = Vec::new();                                               // sp1.rs:128-130 / prover.rs:101-103
        }
        Ok((output, blob))
    }
}

fn prove_one(ctx: &Context, d_trace: *const u32, log_n: i32, width: u32, column_major: bool, pv: &[u32], prm: &ZkhipParams) -> Result<Vec<u8>> {
    let cap = unsafe { ffi::zkhip_proof_size(log_n, width, prm, pv.len()) };
    anyhow::ensure!(cap > 0, "bad shard shape");
    let mut proof = vec![0u8; cap];
    let mut len = 0usize;
    let rc = unsafe {
        if column_major {
            ffi::zkhip_prove_segment(ctx.raw(), d_trace, log_n, width, pv.as_ptr(), pv.len(), prm, proof.as_mut_ptr(), cap, &mut len)
        } else {
            ffi::zkhip_prove_shard(ctx.raw(), d_trace, width as usize, log_n, width, pv.as_ptr(), pv.len(), prm, proof.as_mut_ptr(), cap, &mut len)
        }
    };
    check(rc, "zkhip_prove")?;
    proof.truncate(len);
    let mut reason = 0;
    check(unsafe { ffi::zkhip_verify_shard(proof.as_ptr(), proof.len(), log_n, width, pv.as_ptr(), pv.len(), prm, &mut reason) }, "zkhip_verify_shard")?;   // sp1.rs:120
    Ok(proof)
}

/// The compress stage (`client.prove(.., Groth16)`, sp1.rs:116: core -> COMPRESS; prover.rs:90: lift -> join): ONE proof that verifies all
/// shard proofs of an execution in-circuit.  `public_values`: those of proof 0, then those of proof 1, ... (`n_public` each).  Returns the
/// joined proof and the verifying key of the shape; `verify_compressed` then needs no byte of the shard proofs.  Limits (docs/RECURSION_NEXT.md):
/// version-1 shard proofs of one shape, at most `zkhip_shard_verifier_max_proofs` per join (136 of the headline shape under an outer proof at blowup 2); `compress_tree` joins the joins.
pub fn compress_shards(ctx: &Context, proofs: &[Vec<u8>], log_n: i32, width: u32, public_values: &[u32], n_public: usize, inner: &ZkhipParams,
                       outer: &ZkhipParams) -> Result<(Vec<u8>, [u32; 8])> {
    anyhow::ensure!(!proofs.is_empty() && public_values.len() == proofs.len() * n_public, "compress_shards: one public-value list per proof");
    let n = proofs.len();
    let mut key: *mut ffi::ZkhipMachineKey = std::ptr::null_mut();
    let mut vk = [0u32; 8];
    check(unsafe { ffi::zkhip_shard_verifier_setup(ctx.raw(), log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, n, outer, &mut key, vk.as_mut_ptr()) },
          "zkhip_shard_verifier_setup")?;
    let cap = unsafe { ffi::zkhip_shard_verifier_proof_size(log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, n, outer) };
    let ptrs: Vec<*const u8> = proofs.iter().map(|p| p.as_ptr()).collect();
    let lens: Vec<usize> = proofs.iter().map(|p| p.len()).collect();
    let mut out = vec![0u8; cap];
    let mut len = 0usize;
    let rc = unsafe {
        ffi::zkhip_prove_shard_verifier(ctx.raw(), key, ptrs.as_ptr(), lens.as_ptr(), n, log_n, width, public_values.as_ptr(), n_public, inner, outer, out.as_mut_ptr(), cap, &mut len)
    };
    unsafe { ffi::zkhip_machine_key_destroy(key) };
    check(rc, "zkhip_prove_shard_verifier")?;
    out.truncate(len);
    Ok((out, vk))
}

/// The compress stage for an execution of ANY shard count (the C++ mirror's `HipGuestProver::with_compress`): while the shards fit one join
/// (`zkhip_shard_verifier_max_proofs`: 136 proofs of the headline shape under an outer proof at blowup 2) that is one call of `compress_shards`; beyond, `ceil(n / max)` joins of
/// equal size -- the last one repeats the last shard proof to fill its places, so that every join has the same shape and ONE key.  Returns the
/// joined proofs in shard order, the key and the join size J (a verifier rebuilds the public-value lists from it the same way).
pub fn compress_execution(ctx: &Context, proofs: &[Vec<u8>], log_n: i32, width: u32, public_values: &[u32], n_public: usize, inner: &ZkhipParams,
                          outer: &ZkhipParams) -> Result<(Vec<Vec<u8>>, [u32; 8], usize)> {
    anyhow::ensure!(!proofs.is_empty() && public_values.len() == proofs.len() * n_public, "compress_execution: one public-value list per proof");
    let n = proofs.len();
    let most = unsafe { ffi::zkhip_shard_verifier_max_proofs(log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, outer) };
    anyhow::ensure!(most > 0, "compress_execution: the shard verifier does not take this shape");
    let joins = (n + most - 1) / most;
    let j = (n + joins - 1) / joins;
    let (mut out, mut vk) = (Vec::with_capacity(joins), [0u32; 8]);
    for c in 0..joins {
        let idx: Vec<usize> = (0..j).map(|k| std::cmp::min(c * j + k, n - 1)).collect();
        let chunk: Vec<Vec<u8>> = idx.iter().map(|&i| proofs[i].clone()).collect();
        let pvs: Vec<u32> = idx.iter().flat_map(|&i| public_values[i * n_public..(i + 1) * n_public].iter().copied()).collect();
        let (joined, key) = compress_shards(ctx, &chunk, log_n, width, &pvs, n_public, inner, outer)?;
        anyhow::ensure!(c == 0 || key == vk, "compress_execution: joins of one shape must share a key");
        vk = key;
        out.push(joined);
    }
    Ok((out, vk, j))
}

/// The join machine over `join` shard proofs of a shape, as the inner machine of machine mode (zkhip_machine_desc): the library's own
/// description of its eight chips + the join key's root.  The vectors own the words the raw description points into.
pub struct JoinMachine {
    progs: Vec<Vec<u32>>, tabs: Vec<Vec<u32>>, pp: Vec<*const u32>, tp: Vec<*const u32>, pw: Vec<usize>, tw: Vec<usize>,
    lns: [i32; 8], widths: [u32; 8], pres: [u32; 8],
    pub desc: ffi::ZkhipMachineDesc,
}
impl JoinMachine {
    pub fn new(log_n: i32, width: u32, n_public: usize, join: usize, join_key: &[u32; 8], outer_of_the_joins: &ZkhipParams, inner: &ZkhipParams) -> Result<Box<Self>> {
        let mut m = Box::new(JoinMachine { progs: vec![], tabs: vec![], pp: vec![], tp: vec![], pw: vec![], tw: vec![], lns: [0; 8], widths: [0; 8], pres: [0; 8],
                                           desc: unsafe { std::mem::zeroed() } });
        for i in 0..8 {
            let (mut ln, mut mw, mut pw) = (0i32, 0u32, 0u32);
            for kind in 0..2 {
                let n = unsafe { ffi::zkhip_shard_verifier_describe(log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, join, i, kind, std::ptr::null_mut(), 0, &mut ln, &mut mw, &mut pw) };
                anyhow::ensure!(n > 0, "zkhip_shard_verifier_describe");
                let mut v = vec![0u32; n];
                unsafe { ffi::zkhip_shard_verifier_describe(log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, join, i, kind, v.as_mut_ptr(), n, &mut ln, &mut mw, &mut pw) };
                if kind == 0 { m.progs.push(v) } else { m.tabs.push(v) }
            }
            m.lns[i as usize] = ln; m.widths[i as usize] = mw; m.pres[i as usize] = pw;
        }
        m.pp = m.progs.iter().map(|v| v.as_ptr()).collect(); m.tp = m.tabs.iter().map(|v| v.as_ptr()).collect();
        m.pw = m.progs.iter().map(|v| v.len()).collect(); m.tw = m.tabs.iter().map(|v| v.len()).collect();
        m.desc = ffi::ZkhipMachineDesc { n_chips: 8, log_ns: m.lns.as_ptr(), widths: m.widths.as_ptr(), pre_widths: m.pres.as_ptr(), programs: m.pp.as_ptr(), program_words: m.pw.as_ptr(),
                                         tables: m.tp.as_ptr(), table_words: m.tw.as_ptr(), key_root: *join_key, num_queries: outer_of_the_joins.num_queries,
                                         pow_bits: outer_of_the_joins.pow_bits, n_public: (n_public * join) as u32 };
        Ok(m)
    }
}

/// The tree's first level in ONE call: `proofs.len() / j` joins of one shape, dealt over `devices` (empty: every visible one), `in_flight` at a time on
/// each, every join checked on its worker's thread (`zkhip_prove_shard_verifier_batch`).  Returns the joins and the key of the shape.
pub fn join_batch(devices: &[i32], proofs: &[Vec<u8>], j: usize, log_n: i32, width: u32, public_values: &[u32], n_public: usize, inner: &ZkhipParams, outer: &ZkhipParams,
                  in_flight: i32) -> Result<(Vec<Vec<u8>>, [u32; 8])> {
    anyhow::ensure!(j >= 1 && !proofs.is_empty() && proofs.len() % j == 0 && public_values.len() == proofs.len() * n_public, "join_batch: a multiple of j shard proofs, n_public values each");
    let n_joins = proofs.len() / j;
    let cap = unsafe { ffi::zkhip_shard_verifier_proof_size(log_n, width, inner.num_queries as usize, inner.pow_bits, n_public, j, outer) };
    anyhow::ensure!(cap != 0, "join_batch: bad shape");
    let ptrs: Vec<*const u8> = proofs.iter().map(|p| p.as_ptr()).collect();
    let lens: Vec<usize> = proofs.iter().map(|p| p.len()).collect();
    let mut out = vec![0u8; n_joins * cap];
    let mut out_lens = vec![0usize; n_joins];
    let mut vk = [0u32; 8];
    check(unsafe {
        ffi::zkhip_prove_shard_verifier_batch(if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() }, devices.len() as i32, ptrs.as_ptr(), lens.as_ptr(), proofs.len(), j,
                                              log_n, width, public_values.as_ptr(), n_public, inner, outer, in_flight, 1, out.as_mut_ptr(), cap, out_lens.as_mut_ptr(), vk.as_mut_ptr())
    }, "zkhip_prove_shard_verifier_batch")?;
    Ok(((0..n_joins).map(|c| out[c * cap..c * cap + out_lens[c]].to_vec()).collect(), vk))
}

/// THE TREE (sp1-recursion joins the joins; prover.rs:90 lift -> join): `joins` are the proofs of `compress_execution` (join size `j`, key `join_vk`);
/// ONE proof verifies them all in-circuit (machine mode of the shard verifier machine).  Returns that proof and its key (a verifier derives the
/// same key with `zkhip_machine_verifier_key_host`).  `public_values`: the shard proofs' in join order (`j * n_public` per join).
pub fn compress_tree(ctx: &Context, joins: &[Vec<u8>], log_n: i32, width: u32, public_values: &[u32], n_public: usize, j: usize, join_vk: &[u32; 8], inner: &ZkhipParams,
                     outer: &ZkhipParams) -> Result<(Vec<u8>, [u32; 8])> {
    anyhow::ensure!(joins.len() >= 2 && public_values.len() == joins.len() * j * n_public, "compress_tree: j * n_public public values per join");
    let jm = JoinMachine::new(log_n, width, n_public, j, join_vk, outer, inner)?;
    let n = joins.len();
    let mut key: *mut ffi::ZkhipMachineKey = std::ptr::null_mut();
    let mut vk = [0u32; 8];
    check(unsafe { ffi::zkhip_machine_verifier_setup(ctx.raw(), &jm.desc, n, outer, &mut key, vk.as_mut_ptr()) }, "zkhip_machine_verifier_setup")?;
    let cap = unsafe { ffi::zkhip_machine_verifier_proof_size(&jm.desc, n, outer) };
    let ptrs: Vec<*const u8> = joins.iter().map(|p| p.as_ptr()).collect();
    let lens: Vec<usize> = joins.iter().map(|p| p.len()).collect();
    let mut out = vec![0u8; cap];
    let mut len = 0usize;
    let rc = unsafe { ffi::zkhip_prove_machine_verifier(ctx.raw(), key, &jm.desc, ptrs.as_ptr(), lens.as_ptr(), n, public_values.as_ptr(), j * n_public, outer, out.as_mut_ptr(), cap, &mut len) };
    unsafe { ffi::zkhip_machine_key_destroy(key) };
    check(rc, "zkhip_prove_machine_verifier")?;
    out.truncate(len);
    Ok((out, vk))
}

/// Host-only check of a tree's top: the join machine's description (a function of the shard shape and the join key), the shard proofs' public values, the key.
pub fn verify_tree(top: &[u8], log_n: i32, width: u32, public_values: &[u32], n_public: usize, j: usize, n_joins: usize, join_vk: &[u32; 8], inner: &ZkhipParams,
                   outer: &ZkhipParams) -> Result<()> {
    let jm = JoinMachine::new(log_n, width, n_public, j, join_vk, outer, inner)?;
    let mut vk = [0u32; 8];
    check(unsafe { ffi::zkhip_machine_verifier_key_host(&jm.desc, n_joins, outer, vk.as_mut_ptr()) }, "zkhip_machine_verifier_key_host")?;
    let mut reason = 0;
    check(unsafe { ffi::zkhip_verify_machine_recursive(&jm.desc, top.as_ptr(), top.len(), public_values.as_ptr(), j * n_public, n_joins, vk.as_ptr(), outer, &mut reason) },
          "zkhip_verify_machine_recursive")
}

/// Host-only check of a joined proof: the shape, the shard proofs' public values and the key of the shape (sp1.rs:120 for the compressed proof).
pub fn verify_compressed(proof: &[u8], log_n: i32, width: u32, public_values: &[u32], n_public: usize, n_proofs: usize, vk: &[u32; 8], inner: &ZkhipParams,
                         outer: &ZkhipParams) -> Result<()> {
    let mut reason = 0;
    check(unsafe {
        ffi::zkhip_verify_shard_recursive(proof.as_ptr(), proof.len(), log_n, width, inner.num_queries as usize, inner.pow_bits, public_values.as_ptr(), n_public, n_proofs,
                                          vk.as_ptr(), outer, &mut reason)
    }, "zkhip_verify_shard_recursive")
}


// ---- shards in SP1's shard STRUCTURE (the C++ mirror's MachinePlan, zktls_amd/host/guest_prover_hip.hpp): chips of mixed heights, LogUp pairs inside and
// across tables, preprocessed columns committed by setup -- every shard ONE keyed-machine proof (version 11), the compress stage in machine mode ----

/// One chip of a machine shard: `2^log_n x width` columns `[preprocessed | main]` under the synthetic AIR, `pairs` LogUp pairs as an interaction table
/// (in-table when `partner < 0`, else exchanged with the chip `partner` of the same height), `pre_width` leading columns committed once by `machine_setup`.
#[derive(Clone, Copy, Debug)]
pub struct ChipPlan { pub log_n: i32, pub width: u32, pub pairs: u32, pub partner: i32, pub pre_width: u32 }

/// The keyed machine of a plan: programs, interaction tables, heights and widths as the library's entries take them.
pub struct MachineShape {
    pub chips: Vec<ChipPlan>,
    progs: Vec<Vec<u32>>, tabs: Vec<Vec<u32>>, pp: Vec<*const u32>, tp: Vec<*const u32>, pw: Vec<usize>, tw: Vec<usize>,
    lns: Vec<i32>, widths: Vec<u32>, pres: Vec<u32>,
}
const LKUP_MAGIC: u32 = 0x5055_4B4C;
const BUS_SP1: u32 = 300;
const KEY_SHARD: u64 = 9999;
const MACHINE_PUBLICS: usize = 9;             // request digest | shard index

impl MachineShape {
    /// bench.py's `multichip` shard: 2^20 x 96 and 2^20 x 32 looking each other up, 2^19 x 64, 2^18 x 128, 2^16 x 256 (32 preprocessed columns), 2^14 x 40
    pub fn sp1_shaped() -> Result<Box<Self>> {
        Self::new(vec![ChipPlan { log_n: 20, width: 96, pairs: 3, partner: 1, pre_width: 0 }, ChipPlan { log_n: 20, width: 32, pairs: 3, partner: 0, pre_width: 0 },
                       ChipPlan { log_n: 19, width: 64, pairs: 2, partner: -1, pre_width: 0 }, ChipPlan { log_n: 18, width: 128, pairs: 4, partner: -1, pre_width: 0 },
                       ChipPlan { log_n: 16, width: 256, pairs: 8, partner: -1, pre_width: 32 }, ChipPlan { log_n: 14, width: 40, pairs: 1, partner: -1, pre_width: 0 }])
    }
    pub fn new(chips: Vec<ChipPlan>) -> Result<Box<Self>> {
        anyhow::ensure!(!chips.is_empty() && chips.len() <= 16 && chips.iter().any(|c| c.pre_width != 0), "machine plan: 1 to 16 chips, one of them with preprocessed columns");
        let mut m = Box::new(MachineShape { chips, progs: vec![], tabs: vec![], pp: vec![], tp: vec![], pw: vec![], tw: vec![], lns: vec![], widths: vec![], pres: vec![] });
        for (c, ch) in m.chips.clone().iter().enumerate() {
            anyhow::ensure!(c == 0 || ch.log_n <= m.chips[c - 1].log_n, "machine plan: chips tallest first");
            anyhow::ensure!(ch.width % 4 == 0 && 8 * ch.pairs <= ch.width, "machine plan: widths in multiples of 4, 8 columns per LogUp pair");
            anyhow::ensure!(ch.pre_width == 0 || (ch.pre_width % 8 == 0 && ch.pre_width < ch.width && ch.partner < 0), "machine plan: preprocessed columns are whole in-table pairs");
            if ch.partner >= 0 {
                let o = m.chips.get(ch.partner as usize).ok_or_else(|| anyhow!("machine plan: partner out of range"))?;
                anyhow::ensure!(o.partner == c as i32 && o.log_n == ch.log_n && o.pairs == ch.pairs, "machine plan: partners are mutual, of one height and one pair count");
            }
            let mut words = 0usize;
            check(unsafe { ffi::zkhip_air_synthetic(ch.width, MACHINE_PUBLICS, std::ptr::null_mut(), 0, &mut words) }, "zkhip_air_synthetic")?;
            let mut prog = vec![0u32; words];
            check(unsafe { ffi::zkhip_air_synthetic(ch.width, MACHINE_PUBLICS, prog.as_mut_ptr(), words, &mut words) }, "zkhip_air_synthetic")?;
            let mut tab = vec![LKUP_MAGIC, 2 * ch.pairs, 0];
            for q in 0..ch.pairs {
                let to = if ch.partner < 0 { c as u32 } else { ch.partner as u32 };
                tab.extend_from_slice(&[0, u32::MAX, BUS_SP1 + 16 * c as u32 + q, 2, 8 * q, 8 * q + 1]);            // send (a, b) of group 2q
                tab.extend_from_slice(&[1, u32::MAX, BUS_SP1 + 16 * to + q, 2, 8 * q + 4, 8 * q + 5]);               // receive at group 2q + 1
            }
            tab[2] = tab.len() as u32;
            m.progs.push(prog);
            m.tabs.push(if ch.pairs > 0 { tab } else { vec![] });
            m.lns.push(ch.log_n); m.widths.push(ch.width - ch.pre_width); m.pres.push(ch.pre_width);
        }
        m.pp = m.progs.iter().map(|v| v.as_ptr()).collect();
        m.tp = m.tabs.iter().map(|v| if v.is_empty() { std::ptr::null() } else { v.as_ptr() }).collect();
        m.pw = m.progs.iter().map(|v| v.len()).collect(); m.tw = m.tabs.iter().map(|v| v.len()).collect();
        Ok(m)
    }
    /// the machine as the INNER machine of machine mode (zkhip_machine_desc): `key_root` from `machine_setup`
    pub fn desc(&self, key_root: &[u32; 8], prm: &ZkhipParams) -> ffi::ZkhipMachineDesc {
        ffi::ZkhipMachineDesc { n_chips: self.chips.len() as i32, log_ns: self.lns.as_ptr(), widths: self.widths.as_ptr(), pre_widths: self.pres.as_ptr(), programs: self.pp.as_ptr(),
                                program_words: self.pw.as_ptr(), tables: self.tp.as_ptr(), table_words: self.tw.as_ptr(), key_root: *key_root, num_queries: prm.num_queries,
                                pow_bits: prm.pow_bits, n_public: MACHINE_PUBLICS as u32 }
    }
}

fn stream_seed(digest: &[u32; 8]) -> u64 { digest[..4].iter().fold(0u64, |h, w| (h << 16) ^ (*w as u64)) }

/// `client.setup(elf)` (sp1.rs:113) for a machine plan: the preprocessed columns -- a function of the PROGRAM -- committed on `ctx`.  The key belongs to `ctx`.
pub fn machine_setup(ctx: &Context, shape: &MachineShape, elf: &[u8], prm: &ZkhipParams) -> Result<(*mut ffi::ZkhipMachineKey, [u32; 8])> {
    let mut pd = [0u32; 8];
    check(unsafe { ffi::zkhip_request_digest(std::ptr::null(), 0, elf.as_ptr(), elf.len(), pd.as_mut_ptr()) }, "zkhip_request_digest")?;
    let seed = stream_seed(&pd);
    let mut bufs = Vec::new();
    let mut pre = Vec::new();
    for (c, ch) in shape.chips.iter().enumerate() {
        if ch.pre_width == 0 {
            pre.push(ffi::ZkhipChip { d_trace: std::ptr::null(), ld: 0, log_n: ch.log_n, width: 0, logup_pairs: 0, partner: -1 });
            continue;
        }
        let buf = ctx.alloc((ch.width as usize) << ch.log_n)?;
        check(unsafe { ffi::zkhip_gen_trace_logup(ctx.raw(), seed, 100 * KEY_SHARD + c as u64, ch.log_n, ch.width, ch.pairs as i32, buf.ptr, ch.width as usize) }, "zkhip_gen_trace_logup")?;
        pre.push(ffi::ZkhipChip { d_trace: buf.ptr, ld: ch.width as usize, log_n: ch.log_n, width: ch.pre_width, logup_pairs: 0, partner: -1 });
        bufs.push(buf);
    }
    let (mut key, mut root) = (std::ptr::null_mut(), [0u32; 8]);
    check(unsafe { ffi::zkhip_machine_setup(ctx.raw(), pre.as_ptr(), pre.len() as i32, prm, &mut key, root.as_mut_ptr()) }, "zkhip_machine_setup")?;
    Ok((key, root))                              // (the key keeps its own device copies: `bufs` may go)
}

/// Shard `s` of the request `digest` as ONE keyed-machine proof, checked against the key like sp1.rs:120.  Traces: the C++ mirror's streams (seed from the request
/// digest, stream 100 s + chip), so both hosts make the same bytes.
pub fn prove_machine_shard(ctx: &Context, shape: &MachineShape, key: *const ffi::ZkhipMachineKey, root: &[u32; 8], digest: &[u32; 8], s: u32, prm: &ZkhipParams) -> Result<Vec<u8>> {
    let seed = stream_seed(digest);
    let mut bufs = Vec::new();
    let mut chips = Vec::new();
    for (c, ch) in shape.chips.iter().enumerate() {
        let buf = ctx.alloc((ch.width as usize) << ch.log_n)?;
        let stream = 100 * s as u64 + c as u64;
        let rc = if ch.partner < 0 {
            unsafe { ffi::zkhip_gen_trace_logup(ctx.raw(), seed, stream, ch.log_n, ch.width, ch.pairs as i32, buf.ptr, ch.width as usize) }
        } else {
            let o = &shape.chips[ch.partner as usize];
            unsafe { ffi::zkhip_gen_trace_logup_cross(ctx.raw(), seed, stream, 100 * s as u64 + ch.partner as u64, ch.log_n, ch.width, o.width, ch.pairs as i32, buf.ptr, ch.width as usize) }
        };
        check(rc, "zkhip_gen_trace_logup")?;
        chips.push(ffi::ZkhipChip { d_trace: unsafe { buf.ptr.add(ch.pre_width as usize) }, ld: ch.width as usize, log_n: ch.log_n, width: ch.width - ch.pre_width, logup_pairs: 0, partner: -1 });
        bufs.push(buf);
    }
    let mut pv = digest.to_vec();
    pv.push(s);
    let n = shape.chips.len() as i32;
    let cap = unsafe { ffi::zkhip_machine_proof_size_keyed(shape.lns.as_ptr(), shape.widths.as_ptr(), shape.pres.as_ptr(), shape.pp.as_ptr(), shape.pw.as_ptr(), shape.tp.as_ptr(), shape.tw.as_ptr(), n, prm, pv.len()) };
    anyhow::ensure!(cap > 0, "bad machine plan");
    let (mut proof, mut len) = (vec![0u8; cap], 0usize);
    check(unsafe { ffi::zkhip_prove_machine_keyed(ctx.raw(), key, chips.as_ptr(), shape.pp.as_ptr(), shape.pw.as_ptr(), shape.tp.as_ptr(), shape.tw.as_ptr(), n, pv.as_ptr(), pv.len(), prm,
                                                  proof.as_mut_ptr(), cap, &mut len) }, "zkhip_prove_machine_keyed")?;
    proof.truncate(len);
    let mut reason = 0;
    check(unsafe { ffi::zkhip_verify_machine_keyed(proof.as_ptr(), proof.len(), shape.lns.as_ptr(), shape.widths.as_ptr(), shape.pres.as_ptr(), root.as_ptr(), shape.pp.as_ptr(), shape.pw.as_ptr(),
                                                   shape.tp.as_ptr(), shape.tw.as_ptr(), n, pv.as_ptr(), pv.len(), prm, &mut reason) }, "zkhip_verify_machine_keyed")?;   // sp1.rs:120
    Ok(proof)
}

/// core -> compress for machine shards (sp1.rs:116): ONE machine-mode proof over `proofs` (at most 64, or a Poseidon2 chip of 2^22 rows).  `public_values`: 9 per proof.
/// Returns the proof and its key; `verify_machine_compressed` derives the same key on the host.
pub fn compress_machine_shards(ctx: &Context, shape: &MachineShape, root: &[u32; 8], proofs: &[Vec<u8>], public_values: &[u32], prm: &ZkhipParams) -> Result<(Vec<u8>, [u32; 8])> {
    anyhow::ensure!(!proofs.is_empty() && public_values.len() == proofs.len() * MACHINE_PUBLICS, "compress_machine_shards: 9 public values per proof");
    let desc = shape.desc(root, prm);
    let n = proofs.len();
    let cap = unsafe { ffi::zkhip_machine_verifier_proof_size(&desc, n, prm) };
    anyhow::ensure!(cap > 0, "compress_machine_shards: more proofs than one join takes");
    let (mut key, mut vk) = (std::ptr::null_mut(), [0u32; 8]);
    check(unsafe { ffi::zkhip_machine_verifier_setup(ctx.raw(), &desc, n, prm, &mut key, vk.as_mut_ptr()) }, "zkhip_machine_verifier_setup")?;
    let ptrs: Vec<*const u8> = proofs.iter().map(|p| p.as_ptr()).collect();
    let lens: Vec<usize> = proofs.iter().map(|p| p.len()).collect();
    let (mut out, mut len) = (vec![0u8; cap], 0usize);
    let rc = unsafe { ffi::zkhip_prove_machine_verifier(ctx.raw(), key, &desc, ptrs.as_ptr(), lens.as_ptr(), n, public_values.as_ptr(), MACHINE_PUBLICS, prm, out.as_mut_ptr(), cap, &mut len) };
    unsafe { ffi::zkhip_machine_key_destroy(key) };
    check(rc, "zkhip_prove_machine_verifier")?;
    out.truncate(len);
    Ok((out, vk))
}

/// Host-only check of `compress_machine_shards`' proof from (the plan, the machine key's root, the shards' public values): the join's key is derived here.
pub fn verify_machine_compressed(shape: &MachineShape, root: &[u32; 8], proof: &[u8], public_values: &[u32], n_proofs: usize, prm: &ZkhipParams) -> Result<()> {
    let desc = shape.desc(root, prm);
    let mut vk = [0u32; 8];
    check(unsafe { ffi::zkhip_machine_verifier_key_host(&desc, n_proofs, prm, vk.as_mut_ptr()) }, "zkhip_machine_verifier_key_host")?;
    let mut reason = 0;
    check(unsafe { ffi::zkhip_verify_machine_recursive(&desc, proof.as_ptr(), proof.len(), public_values.as_ptr(), MACHINE_PUBLICS, n_proofs, vk.as_ptr(), prm, &mut reason) },
          "zkhip_verify_machine_recursive")
}
