//! Safe wrapper: a Groth16 circuit resident on one MI355X, proving through `cg_prove`.
//!
//! `GpuCircuit::create_proof` has the semantics of
//! `Groth16::<Bn254, LibsnarkReduction>::create_proof_with_reduction_and_matrices`
//! (forks/groth16/src/prover.rs:26-51): same operands, same `Proof`, byte for byte.
pub mod sys;

use ark_bn254::{Bn254, Fq, Fq2, Fr, G1Affine, G2Affine};
use ark_ff::{BigInteger, PrimeField};
use ark_groth16::r1cs_to_qap::{LibsnarkReduction, R1CSToQAP};
use ark_groth16::{Proof, ProvingKey};
use ark_poly::EvaluationDomain;
use ark_relations::r1cs::{ConstraintMatrices, ConstraintSystemRef, SynthesisError};
use ark_serialize::CanonicalDeserialize;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::{Arc, Mutex, OnceLock};

// `Affine { x, y, infinity }` is repr(Rust): coordinates are packed into byte arrays, struct pointers never
// cross the boundary.  The Montgomery limbs are copied as they are (`Fp.0` is the BigInt of x·2^256 mod p, the
// form forks/circom-compat/src/zkey.rs:397-402 pins), so no field arithmetic happens on this side.
fn put_fq(out: &mut [u8], x: &Fq) {
    for (k, limb) in x.0 .0.iter().enumerate() {
        out[k * 8..k * 8 + 8].copy_from_slice(&limb.to_le_bytes());
    }
}
fn put_fq2(out: &mut [u8], x: &Fq2) {
    put_fq(&mut out[..32], &x.c0);
    put_fq(&mut out[32..64], &x.c1);
}
fn pack_g1(v: &[G1Affine]) -> Vec<u8> {
    let mut out = vec![0u8; v.len() * 64];
    for (i, p) in v.iter().enumerate() {
        if p.infinity {
            continue; // identity = 64 zero bytes
        }
        put_fq(&mut out[i * 64..i * 64 + 32], &p.x);
        put_fq(&mut out[i * 64 + 32..i * 64 + 64], &p.y);
    }
    out
}
fn pack_g2(v: &[G2Affine]) -> Vec<u8> {
    let mut out = vec![0u8; v.len() * 128];
    for (i, p) in v.iter().enumerate() {
        if p.infinity {
            continue; // identity = 128 zero bytes
        }
        put_fq2(&mut out[i * 128..i * 128 + 64], &p.x); // x.c0 ‖ x.c1
        put_fq2(&mut out[i * 128 + 64..i * 128 + 128], &p.y); // y.c0 ‖ y.c1
    }
    out
}

struct Csr {
    row_ptr: Vec<u64>,
    col: Vec<u32>,
    coeff: Vec<u8>,
}
fn to_csr<F: PrimeField>(rows: &[Vec<(F, usize)>]) -> Csr {
    let mut m = Csr { row_ptr: vec![0u64], col: Vec::new(), coeff: Vec::new() };
    for row in rows {
        for (c, j) in row {
            m.col.push(*j as u32);
            m.coeff.extend_from_slice(&c.into_bigint().to_bytes_le()); // canonical, 32 bytes
        }
        m.row_ptr.push(m.col.len() as u64);
    }
    m
}
impl Csr {
    fn view(&self) -> sys::cg_csr {
        sys::cg_csr { row_ptr: self.row_ptr.as_ptr(), col: self.col.as_ptr(), coeff: self.coeff.as_ptr(), nnz: self.col.len() as u64 }
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::cg_last_error()).to_string_lossy().into_owned() }
}
fn map_err(rc: i32) -> SynthesisError {
    eprintln!("crescent-gpu: {} (code {})", last_error(), rc);
    match rc {
        sys::CG_ERR_POLY_DEGREE_TOO_LARGE => SynthesisError::PolynomialDegreeTooLarge, // r1cs_to_qap.rs:156-157
        sys::CG_ERR_MALFORMED_KEY => SynthesisError::MalformedVerifyingKey,
        _ => SynthesisError::AssignmentMissing, // no twin in SynthesisError; the message is on stderr
    }
}

/// Page-locked host bytes from `cg_host_alloc`, freed with `cg_host_free`.
struct PinnedBytes {
    ptr: *mut u8,
    len: usize,
}
impl PinnedBytes {
    /// None when the page-lock fails (the caller then falls back to pageable memory, which `cg_prove` also takes)
    fn new(len: usize) -> Option<Self> {
        let p = unsafe { sys::cg_host_alloc(len as u64) } as *mut u8;
        if p.is_null() {
            eprintln!("crescent-gpu: page-locked witness buffer unavailable ({}): using pageable memory", last_error());
            return None;
        }
        Some(PinnedBytes { ptr: p, len })
    }
    #[allow(clippy::mut_from_ref)]
    fn slice_mut(&self) -> &mut [u8] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}
impl Drop for PinnedBytes {
    fn drop(&mut self) {
        unsafe { sys::cg_host_free(self.ptr as *mut std::os::raw::c_void) }
    }
}

// PinnedBytes is only ever touched by the thread that took it out of the pool
unsafe impl Send for PinnedBytes {}

pub struct GpuCircuit {
    ctx: *mut sys::cg_ctx,
    num_variables: usize,
    /// Page-locked witness buffers, kept for the life of the circuit: `hipHostMalloc` page-locks 48 MB at full size (a few
    /// milliseconds) and `hipHostFree` synchronises the WHOLE device, so allocating and freeing one per proof would stall
    /// every finishing caller behind every proof in flight.  A proof takes a buffer out and puts it back; at most
    /// `pool_max` (= proof_slots + 2, the upload buffers the context holds) are retained, all freed in `Drop`.
    pool: Mutex<Vec<PinnedBytes>>,
    pool_max: usize,
}
// calls on one context are multiplexed over its proof slots inside the library
unsafe impl Send for GpuCircuit {}
unsafe impl Sync for GpuCircuit {}

impl GpuCircuit {
    /// One-time: pack the key and the matrices and copy them to the GPU (`cg_circuit_load`).
    /// `proof_slots` = proofs that may be in flight on this circuit at once (one per calling thread).
    /// The load is STAGED (`CG_FLAG_STAGED_LOAD`): it returns as soon as the circuit can prove and the library finishes its
    /// tables behind the first proofs - what `create_client_state` (creds/src/lib.rs:255-301), which loads and proves once,
    /// wants.  A host that is about to stream proofs may call `wait_ready` first.
    pub fn load(pk: &ProvingKey<Bn254>, m: &ConstraintMatrices<Fr>, device: i32, proof_slots: i32) -> Result<Self, SynthesisError> {
        Self::load_with_flags(pk, m, device, proof_slots, sys::CG_FLAG_STAGED_LOAD)
    }

    /// `load` with the context's flags spelt out (`sys::CG_FLAG_*`; 0 = the synchronous load).
    pub fn load_with_flags(pk: &ProvingKey<Bn254>, m: &ConstraintMatrices<Fr>, device: i32, proof_slots: i32, flags: i32)
                           -> Result<Self, SynthesisError> {
        Self::load_opt(pk, m, sys::cg_options { device, proof_slots, flags, ..Default::default() })
    }

    /// One shard of a proof split over `shard_count` GPUs (`cg_options.shard_rank / shard_count`): the context
    /// owns its ranges of the five queries and answers `prove_partial` (or, loaded with `CG_FLAG_H_SCALARS_EXTERNAL`,
    /// `prove_partial_q` only).  `assemble` on any shard finishes the gathered partial sums into the proof.
    pub fn load_shard(pk: &ProvingKey<Bn254>, m: &ConstraintMatrices<Fr>, device: i32, proof_slots: i32, shard_rank: i32,
                      shard_count: i32, flags: i32) -> Result<Self, SynthesisError> {
        Self::load_opt(pk, m, sys::cg_options { device, proof_slots, shard_rank, shard_count, flags, ..Default::default() })
    }

    /// `load_shard` with an UNEQUAL share (`cg_options.shard_span`): the shard owns `[n·lo/10000, n·hi/10000)` of every query.
    /// The spans of a proof's shards must tile `[0, 10000]`; the ranks that also compute (half of) the witness map take the
    /// smaller ones.
    pub fn load_shard_span(pk: &ProvingKey<Bn254>, m: &ConstraintMatrices<Fr>, device: i32, proof_slots: i32, shard_rank: i32,
                           shard_count: i32, span: (u16, u16), flags: i32) -> Result<Self, SynthesisError> {
        let shard_span = span.0 as i32 | ((span.1 as i32) << 16);
        Self::load_opt(pk, m, sys::cg_options { device, proof_slots, shard_rank, shard_count, flags, shard_span, ..Default::default() })
    }

    fn load_opt(pk: &ProvingKey<Bn254>, m: &ConstraintMatrices<Fr>, opt: sys::cg_options) -> Result<Self, SynthesisError> {
        let proof_slots = opt.proof_slots;
        let rc = unsafe { sys::cg_init(0, std::ptr::null()) };
        if rc != 0 {
            return Err(map_err(rc));
        }
        let (alpha, beta1, delta1) = (pack_g1(&[pk.vk.alpha_g1]), pack_g1(&[pk.beta_g1]), pack_g1(&[pk.delta_g1]));
        let (beta2, delta2) = (pack_g2(&[pk.vk.beta_g2]), pack_g2(&[pk.vk.delta_g2]));
        let (a, b1, h, l) = (pack_g1(&pk.a_query), pack_g1(&pk.b_g1_query), pack_g1(&pk.h_query), pack_g1(&pk.l_query));
        let b2 = pack_g2(&pk.b_g2_query);
        let view = sys::cg_proving_key {
            coord_form: sys::CG_FORM_MONTGOMERY,
            alpha_g1: alpha.as_ptr(),
            beta_g1: beta1.as_ptr(),
            delta_g1: delta1.as_ptr(),
            beta_g2: beta2.as_ptr(),
            delta_g2: delta2.as_ptr(),
            a_query: a.as_ptr(),
            a_len: pk.a_query.len() as u64,
            b_g1_query: b1.as_ptr(),
            b_g1_len: pk.b_g1_query.len() as u64,
            b_g2_query: b2.as_ptr(),
            b_g2_len: pk.b_g2_query.len() as u64,
            h_query: h.as_ptr(),
            h_len: pk.h_query.len() as u64,
            l_query: l.as_ptr(),
            l_len: pk.l_query.len() as u64,
        };
        let (ca, cb, cc) = (to_csr(&m.a), to_csr(&m.b), to_csr(&m.c));
        let abc = [ca.view(), cb.view(), cc.view()];
        let num_variables = m.num_instance_variables + m.num_witness_variables;
        let mut ctx: *mut sys::cg_ctx = std::ptr::null_mut();
        let rc = unsafe {
            sys::cg_circuit_load(&mut ctx, &view, abc.as_ptr(), m.num_instance_variables as u64, m.num_constraints as u64,
                                 num_variables as u64, &opt)
        };
        if rc != 0 {
            return Err(map_err(rc));
        }
        Ok(GpuCircuit { ctx, num_variables, pool: Mutex::new(Vec::new()), pool_max: proof_slots.max(1) as usize + 2 })
    }

    /// forks/groth16/src/prover.rs:26-51 with the key and the matrices already resident on the GPU.
    /// `full_assignment` = instance assignment ‖ witness assignment (element 0 is the constant one).
    pub fn create_proof(&self, r: Fr, s: Fr, full_assignment: &[Fr]) -> Result<Proof<Bn254>, SynthesisError> {
        if full_assignment.len() != self.num_variables {
            return Err(SynthesisError::AssignmentMissing);
        }
        // the canonical bytes are written straight into page-locked memory from the circuit's pool: cg_prove's upload is
        // then one asynchronous DMA at PCIe speed that overlaps the other proofs in flight.  Should page-locking fail
        // (locked-memory limit of the host), a pageable Vec does the same job a little slower - never an error.
        let len = full_assignment.len() * 32;
        let pinned = self.pool.lock().unwrap().pop().or_else(|| PinnedBytes::new(len));
        let mut pageable: Vec<u8> = Vec::new();
        let buf: &mut [u8] = match &pinned {
            Some(p) => p.slice_mut(),
            None => {
                pageable.resize(len, 0);
                &mut pageable[..]
            }
        };
        for (i, x) in full_assignment.iter().enumerate() {
            buf[i * 32..i * 32 + 32].copy_from_slice(&x.into_bigint().to_bytes_le()); // prover.rs:64,71,86 take the same form
        }
        let (rb, sb) = (r.into_bigint().to_bytes_le(), s.into_bigint().to_bytes_le());
        let mut out = [0u8; 256];
        // the phases the reference prints under its `print-trace` feature (forks/groth16/Cargo.toml:48; start_timer! /
        // end_timer! at prover.rs:35-36,62,93,103,115,123), with the GPU's own times: cg_timings mirrors them 1:1
        let mut tm = sys::cg_timings::default();
        let tm_ptr: *mut sys::cg_timings = if cfg!(feature = "print-trace") { &mut tm } else { std::ptr::null_mut() };
        let rc = unsafe { sys::cg_prove(self.ctx, buf.as_ptr(), rb.as_ptr(), sb.as_ptr(), out.as_mut_ptr(), tm_ptr) };
        #[cfg(feature = "print-trace")]
        if rc == 0 {
            print_trace(&tm);
        }
        if let Some(p) = pinned {
            let mut pool = self.pool.lock().unwrap();
            if pool.len() < self.pool_max {
                pool.push(p); // kept for the next proof; a surplus buffer (more callers than slots + 2) is freed here
            }
        }
        if rc != 0 {
            return Err(map_err(rc));
        }
        // a ‖ b ‖ c, ark-serialize uncompressed (data_structures.rs:7-14)
        Proof::deserialize_uncompressed_unchecked(&out[..]).map_err(|_| SynthesisError::MalformedVerifyingKey)
    }
}

/// The reference's `print-trace` output for one proof, phase names as prover.rs spells them, times from cg_timings (the
/// five MSMs overlap on the GPU, so the phases do not add up to the total the way the CPU prover's do).
#[cfg(feature = "print-trace")]
fn print_trace(tm: &sys::cg_timings) {
    let line = |depth: usize, name: &str, ms: f32| println!("{}End:     {} {:.3}ms", "··".repeat(depth), name, ms);
    println!("Start:   Groth16::Prover");                                                  // prover.rs:35
    line(1, "R1CS to QAP witness map", tm.witness_map_ms);                                 // :36
    line(1, "Compute C", tm.msm_h_ms.max(tm.msm_l_ms));                                    // :62  (h_query and l_query MSMs)
    line(1, "Compute A", tm.msm_a_ms);                                                     // :93
    line(1, "Compute B in G1", tm.msm_b1_ms);                                              // :103 (0 when r = 0: skipped, :102-112)
    line(1, "Compute B in G2", tm.msm_b2_ms);                                              // :115
    line(1, "Finish C", tm.finish_ms);                                                     // :123
    line(1, "(upload of the assignment)", tm.upload_ms);
    line(0, "Groth16::Prover", tm.total_ms);                                               // :48
}

fn canonical_bytes(xs: &[Fr]) -> Vec<u8> {
    let mut out = vec![0u8; xs.len() * 32];
    for (i, x) in xs.iter().enumerate() {
        out[i * 32..i * 32 + 32].copy_from_slice(&x.into_bigint().to_bytes_le());
    }
    out
}

/// One proof over several GPUs, on contexts made by `GpuCircuit::load_shard`.  The 384-byte partial records
/// are what the host gathers (one per shard); nothing else crosses between the shards.
impl GpuCircuit {
    /// This shard's five partial sums h ‖ l ‖ a ‖ b1 ‖ b2 (`cg_prove_partial`; prover.rs:66,74,266 over the shard's ranges).
    pub fn prove_partial(&self, r: Fr, full_assignment: &[Fr]) -> Result<[u8; 384], SynthesisError> {
        if full_assignment.len() != self.num_variables {
            return Err(SynthesisError::AssignmentMissing);
        }
        let (w, rb) = (canonical_bytes(full_assignment), r.into_bigint().to_bytes_le());
        let mut out = [0u8; 384];
        let rc = unsafe { sys::cg_prove_partial(self.ctx, w.as_ptr() as *const _, 0, rb.as_ptr(), out.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 { Err(map_err(rc)) } else { Ok(out) }
    }

    /// The gathered partial sums of all shards -> the proof (`cg_assemble`; prover.rs:76-135).
    pub fn assemble(&self, partials: &[[u8; 384]], r: Fr, s: Fr) -> Result<Proof<Bn254>, SynthesisError> {
        let flat: Vec<u8> = partials.iter().flat_map(|p| p.iter().copied()).collect();
        let (rb, sb) = (r.into_bigint().to_bytes_le(), s.into_bigint().to_bytes_le());
        let mut out = [0u8; 256];
        let rc = unsafe { sys::cg_assemble(self.ctx, flat.as_ptr(), partials.len() as u32, rb.as_ptr(), sb.as_ptr(), out.as_mut_ptr()) };
        if rc != 0 {
            return Err(map_err(rc));
        }
        Proof::deserialize_uncompressed_unchecked(&out[..]).map_err(|_| SynthesisError::MalformedVerifyingKey)
    }

    /// SURVEY 8e's other arrangement, step 1 (`cg_witness_map_coset`): the witness map ONCE, on a context loaded without
    /// `CG_FLAG_H_SCALARS_EXTERNAL`; all `domain_size` coset values as 32-byte canonical scalars, laid out shard-major so
    /// that shard p's share is the contiguous range `h_scalars_slice(p)` - what a scatter sends.
    pub fn witness_map_coset(&self, full_assignment: &[Fr]) -> Result<Vec<u8>, SynthesisError> {
        if full_assignment.len() != self.num_variables {
            return Err(SynthesisError::AssignmentMissing);
        }
        let w = canonical_bytes(full_assignment);
        let mut q = vec![0u8; unsafe { sys::cg_domain_size(self.ctx) } as usize * 32];
        let rc = unsafe { sys::cg_witness_map_coset(self.ctx, w.as_ptr() as *const _, 0, q.as_mut_ptr() as *mut _, 0) };
        if rc != 0 { Err(map_err(rc)) } else { Ok(q) }
    }

    /// (offset, count), in scalars, of shard `shard`'s share of `witness_map_coset`'s output (`cg_h_scalars_slice`).
    pub fn h_scalars_slice(&self, shard: u32) -> Result<(u64, u64), SynthesisError> {
        let (mut off, mut cnt) = (0u64, 0u64);
        let rc = unsafe { sys::cg_h_scalars_slice(self.ctx, shard, &mut off, &mut cnt) };
        if rc != 0 { Err(map_err(rc)) } else { Ok((off, cnt)) }
    }

    /// Step 2 (`cg_prove_partial_q`): this shard's partial sums with its slice of the coset values supplied; the only way a
    /// context loaded with `CG_FLAG_H_SCALARS_EXTERNAL` proves.
    pub fn prove_partial_q(&self, r: Fr, full_assignment: &[Fr], q_slice: &[u8]) -> Result<[u8; 384], SynthesisError> {
        if full_assignment.len() != self.num_variables {
            return Err(SynthesisError::AssignmentMissing);
        }
        let (w, rb) = (canonical_bytes(full_assignment), r.into_bigint().to_bytes_le());
        let mut out = [0u8; 384];
        let rc = unsafe {
            sys::cg_prove_partial_q(self.ctx, w.as_ptr() as *const _, 0, q_slice.as_ptr() as *const _, 0, rb.as_ptr(), out.as_mut_ptr(),
                                    std::ptr::null_mut())
        };
        if rc != 0 { Err(map_err(rc)) } else { Ok(out) }
    }
}

/// A sharded proof between `cg_prove_partial_q_begin` and `_finish` / `_finish2`: the shard's l, a, b1, b2 partial sums are
/// queued and running; the h share follows once the slice (or the two halves' slices) has arrived.  Holds one of the circuit's
/// proof slots; dropped unfinished, it is aborted (`cg_prove_partial_q_abort`).
pub struct OpenProof<'a> {
    p: *mut sys::cg_partial,
    circuit: &'a GpuCircuit,
}

impl GpuCircuit {
    /// `cg_prove_partial_q_begin`: open a sharded proof; returns while the assignment-driven sums run.
    pub fn prove_partial_q_begin(&self, r: Fr, full_assignment: &[Fr]) -> Result<OpenProof<'_>, SynthesisError> {
        if full_assignment.len() != self.num_variables {
            return Err(SynthesisError::AssignmentMissing);
        }
        let (w, rb) = (canonical_bytes(full_assignment), r.into_bigint().to_bytes_le());
        let mut p: *mut sys::cg_partial = std::ptr::null_mut();
        // (the library has copied the assignment to the GPU before it returns: `w` may go)
        let rc = unsafe { sys::cg_prove_partial_q_begin(self.ctx, w.as_ptr() as *const _, 0, rb.as_ptr(), &mut p) };
        if rc != 0 { Err(map_err(rc)) } else { Ok(OpenProof { p, circuit: self }) }
    }
}

impl<'a> OpenProof<'a> {
    /// One side of the coset values for this proof's assignment (`cg_partial_witness_map_coset_half`): `which` = 0 the a side,
    /// 1 the b side; `domain_size` plain canonical scalars, laid out like `witness_map_coset`'s output.
    pub fn witness_map_coset_half(&self, which: i32) -> Result<Vec<u8>, SynthesisError> {
        let mut q = vec![0u8; unsafe { sys::cg_domain_size(self.circuit.ctx) } as usize * 32];
        let rc = unsafe { sys::cg_partial_witness_map_coset_half(self.p, which, q.as_mut_ptr() as *mut _, 0) };
        if rc != 0 { Err(map_err(rc)) } else { Ok(q) }
    }

    /// `cg_prove_partial_q_finish`: the h share with this shard's slice of the coset values -> the 384-byte record.
    pub fn finish(mut self, q_slice: &[u8]) -> Result<[u8; 384], SynthesisError> {
        let mut out = [0u8; 384];
        let p = std::mem::replace(&mut self.p, std::ptr::null_mut());      // the call consumes the handle, success or not
        let rc = unsafe { sys::cg_prove_partial_q_finish(p, q_slice.as_ptr() as *const _, 0, out.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 { Err(map_err(rc)) } else { Ok(out) }
    }

    /// `cg_prove_partial_q_finish2`: the same from the shard's slices of BOTH sides; their products are formed on the GPU.
    pub fn finish2(mut self, a_slice: &[u8], b_slice: &[u8]) -> Result<[u8; 384], SynthesisError> {
        let mut out = [0u8; 384];
        let p = std::mem::replace(&mut self.p, std::ptr::null_mut());
        let rc = unsafe {
            sys::cg_prove_partial_q_finish2(p, a_slice.as_ptr() as *const _, b_slice.as_ptr() as *const _, 0, out.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 { Err(map_err(rc)) } else { Ok(out) }
    }
}

impl<'a> Drop for OpenProof<'a> {
    fn drop(&mut self) {
        if !self.p.is_null() {
            unsafe { sys::cg_prove_partial_q_abort(self.p) }
        }
    }
}

impl GpuCircuit {
    /// Blocks until a staged load's final arrangement is in force (`cg_ctx_wait_ready`); `Ok(false)` = the time ran out
    /// first.  `timeout_ms < 0` waits without limit.  Proofs made before that are the same bytes, only slower.
    pub fn wait_ready(&self, timeout_ms: i32) -> Result<bool, SynthesisError> {
        match unsafe { sys::cg_ctx_wait_ready(self.ctx, timeout_ms) } {
            0 => Ok(true),
            1 => Ok(false),
            rc => Err(map_err(rc)),
        }
    }

    /// Where the load's time went (`cg_ctx_get_load_timings`): the counterpart of the reference's "Reading ProverParams" /
    /// "Reading R1CS" timers (creds/src/lib.rs:257,266).
    pub fn load_timings(&self) -> Result<sys::cg_load_timings, SynthesisError> {
        let mut t = sys::cg_load_timings::default();
        let rc = unsafe { sys::cg_ctx_get_load_timings(self.ctx, &mut t) };
        if rc != 0 {
            return Err(map_err(rc));
        }
        Ok(t)
    }

    /// What the circuit occupies on the GPU and how its MSMs are configured (`cg_ctx_get_info`).
    pub fn info(&self) -> Result<sys::cg_ctx_info, SynthesisError> {
        let mut i = sys::cg_ctx_info::default();
        let rc = unsafe { sys::cg_ctx_get_info(self.ctx, &mut i) };
        if rc != 0 {
            return Err(map_err(rc));
        }
        Ok(i)
    }
}

impl Drop for GpuCircuit {
    fn drop(&mut self) {
        unsafe { sys::cg_circuit_free(self.ctx) } // waits for calls still inside the context
        self.pool.lock().unwrap().clear(); // cg_host_free: the one place the pinned buffers are released
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's own plug point: `Groth16<E, QAP: R1CSToQAP = LibsnarkReduction>` (forks/groth16/src/lib.rs:55-57).
// `Groth16::<Bn254, GpuReduction>::prove(..)` keeps the whole of prover.rs - synthesis, arkworks' MSMs - and moves the
// witness map (three sparse products, seven transforms; r1cs_to_qap.rs:150-213) to the GPU through cg_qap_*.  It is the
// smaller of the two integrations (the other being `GpuCircuit`, which also takes the five MSMs); both give the same bytes.
// ---------------------------------------------------------------------------------------------------------------
struct QapHandle(*mut sys::cg_qap_ctx);
unsafe impl Send for QapHandle {}
unsafe impl Sync for QapHandle {} // calls on one handle serialise inside the library
impl Drop for QapHandle {
    fn drop(&mut self) {
        unsafe { sys::cg_qap_free(self.0) }
    }
}

/// Matrices are constants of a credential type but arrive by reference on every call (r1cs_to_qap.rs:150-155), so the
/// resident copy is looked up by a digest of their WHOLE content: the shape, the non-zero counts and two independent
/// 64-bit FNV-1a passes over every term (all four limbs of every coefficient, every column index, every row boundary).
/// One pass over the 17 M terms of a full-size circuit costs a few tens of milliseconds - the reference's own call spends
/// seconds in the transforms it replaces.  (Round 2 sampled every 1024th term: two circuits of the same shape that
/// differed in an unsampled coefficient would have shared a resident copy and produced a wrong h without an error.)
fn fingerprint<F: PrimeField>(m: &ConstraintMatrices<F>) -> [u64; 8] {
    const P: u64 = 0x0000_0100_0000_01b3; // FNV-1a 64 prime
    let (mut h1, mut h2) = (0xcbf2_9ce4_8422_2325u64, 0x8422_2325_cbf2_9ce4u64);
    let mut eat = |v: u64| {
        h1 = (h1 ^ v).wrapping_mul(P);
        h2 = (h2 ^ v.rotate_left(29) ^ 0x9e37_79b9_7f4a_7c15).wrapping_mul(P).rotate_left(5);
    };
    for mat in [&m.a, &m.b, &m.c] {
        for row in mat.iter() {
            eat(0xffff_ffff_0000_0000 | row.len() as u64); // row boundary
            for (c, j) in row {
                for limb in c.into_bigint().as_ref() {
                    eat(*limb);
                }
                eat(*j as u64);
            }
        }
        eat(0x5a5a_5a5a_5a5a_5a5a); // matrix boundary
    }
    [
        m.num_instance_variables as u64, m.num_witness_variables as u64, m.num_constraints as u64,
        m.a_num_non_zero as u64, m.b_num_non_zero as u64, m.c_num_non_zero as u64, h1, h2,
    ]
}

/// At most this many circuits keep a device copy of their matrices; the least recently used one is dropped first.  A
/// dropped entry's `cg_qap_free` runs when its last `Arc` holder (a call still in flight) lets go - never under a caller.
const QAP_CACHE_MAX: usize = 4;
struct QapCache {
    tick: u64,
    entries: HashMap<[u64; 8], (u64, Arc<QapHandle>)>, // key -> (last use, handle)
}
fn qap_for<F: PrimeField>(m: &ConstraintMatrices<F>) -> Result<Arc<QapHandle>, SynthesisError> {
    static CACHE: OnceLock<Mutex<QapCache>> = OnceLock::new();
    let key = fingerprint(m);
    let mut cache = CACHE.get_or_init(|| Mutex::new(QapCache { tick: 0, entries: HashMap::new() })).lock().unwrap();
    cache.tick += 1;
    let now = cache.tick;
    if let Some(e) = cache.entries.get_mut(&key) {
        e.0 = now;
        return Ok(e.1.clone());
    }
    let rc = unsafe { sys::cg_init(0, std::ptr::null()) };
    if rc != 0 {
        return Err(map_err(rc));
    }
    let (ca, cb, cc) = (to_csr(&m.a), to_csr(&m.b), to_csr(&m.c));
    let abc = [ca.view(), cb.view(), cc.view()];
    let mut ctx: *mut sys::cg_qap_ctx = std::ptr::null_mut();
    let rc = unsafe {
        sys::cg_qap_load(&mut ctx, abc.as_ptr(), m.num_instance_variables as u64, m.num_constraints as u64,
                         (m.num_instance_variables + m.num_witness_variables) as u64, -1)
    };
    if rc != 0 {
        return Err(map_err(rc));
    }
    let h = Arc::new(QapHandle(ctx));
    cache.entries.insert(key, (now, h.clone()));
    while cache.entries.len() > QAP_CACHE_MAX {
        let oldest = *cache.entries.iter().min_by_key(|(_, v)| v.0).map(|(k, _)| k).unwrap();
        cache.entries.remove(&oldest); // the Arc drops here, or with the last call still using it
    }
    Ok(h)
}

/// `impl R1CSToQAP` (forks/groth16/src/r1cs_to_qap.rs:49-98).  Usage: `Groth16::<Bn254, GpuReduction>::prove(&pk, circuit, &mut rng)`.
pub struct GpuReduction;

impl R1CSToQAP for GpuReduction {
    // the generator's half is untouched: it runs once per circuit at zksetup time
    fn instance_map_with_evaluation<F: PrimeField, D: EvaluationDomain<F>>(
        cs: ConstraintSystemRef<F>,
        t: &F,
    ) -> Result<(Vec<F>, Vec<F>, Vec<F>, F, usize, usize), SynthesisError> {
        LibsnarkReduction::instance_map_with_evaluation::<F, D>(cs, t)
    }

    fn witness_map_from_matrices<F: PrimeField, D: EvaluationDomain<F>>(
        matrices: &ConstraintMatrices<F>,
        num_inputs: usize,
        num_constraints: usize,
        full_assignment: &[F],
    ) -> Result<Vec<F>, SynthesisError> {
        // The library computes over BN254's scalar field only; any other field keeps the CPU path.
        if F::MODULUS.to_bytes_le() != <Fr as PrimeField>::MODULUS.to_bytes_le() {
            return LibsnarkReduction::witness_map_from_matrices::<F, D>(matrices, num_inputs, num_constraints, full_assignment);
        }
        // F is BN254's Fr (checked by modulus); everything crosses the boundary as canonical bytes, so no cast is needed
        let m = matrices;
        if num_inputs != m.num_instance_variables || num_constraints != m.num_constraints
            || full_assignment.len() != m.num_instance_variables + m.num_witness_variables
        {
            return Err(SynthesisError::AssignmentMissing);
        }
        let h = qap_for(m)?;
        let mut w = Vec::with_capacity(full_assignment.len() * 32);
        for x in full_assignment {
            w.extend_from_slice(&x.into_bigint().to_bytes_le());
        }
        let d = unsafe { sys::cg_qap_domain_size(h.0) } as usize;
        let mut out = vec![0u8; d * 32];
        let rc = unsafe { sys::cg_qap_witness_map(h.0, w.as_ptr() as *const _, 0, out.as_mut_ptr() as *mut _, 0) };
        if rc != 0 {
            return Err(map_err(rc));
        }
        Ok(out.chunks_exact(32).map(F::from_le_bytes_mod_order).collect()) // domain_size coefficients (r1cs_to_qap.rs:212)
    }

    fn h_query_scalars<F: PrimeField, D: EvaluationDomain<F>>(
        max_power: usize,
        t: F,
        zt: F,
        delta_inverse: F,
    ) -> Result<Vec<F>, SynthesisError> {
        LibsnarkReduction::h_query_scalars::<F, D>(max_power, t, zt, delta_inverse)
    }
}

/// `ConstraintMatrices` of a circom R1CS with column = wire id: the matrices `cs.to_matrices()` yields for
/// `CircomCircuit::generate_constraints` (forks/circom-compat/src/circom/circuit.rs:61-86) when the wire mapping
/// is disabled (builder.rs:63-64).  `constraints` is `circom.r1cs.constraints`.
pub fn matrices_of(num_inputs: usize, num_variables: usize,
                   constraints: &[(Vec<(usize, Fr)>, Vec<(usize, Fr)>, Vec<(usize, Fr)>)]) -> ConstraintMatrices<Fr> {
    let conv = |lc: &Vec<(usize, Fr)>| lc.iter().map(|(j, c)| (*c, *j)).collect::<Vec<_>>();
    let a: Vec<_> = constraints.iter().map(|c| conv(&c.0)).collect();
    let b: Vec<_> = constraints.iter().map(|c| conv(&c.1)).collect();
    let c: Vec<_> = constraints.iter().map(|c| conv(&c.2)).collect();
    ConstraintMatrices {
        num_instance_variables: num_inputs,
        num_witness_variables: num_variables - num_inputs,
        num_constraints: constraints.len(),
        a_num_non_zero: a.iter().map(|r| r.len()).sum(),
        b_num_non_zero: b.iter().map(|r| r.len()).sum(),
        c_num_non_zero: c.iter().map(|r| r.len()).sum(),
        a,
        b,
        c,
    }
}

/// `<G1 as VariableBaseMSM>::msm_bigint` for the show-step sized sums (creds/src/utils.rs:124-138): one-shot.
pub fn msm_g1(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Affine, SynthesisError> {
    let b = pack_g1(bases);
    let mut sc = Vec::with_capacity(scalars.len() * 32);
    for x in scalars {
        sc.extend_from_slice(&x.into_bigint().to_bytes_le());
    }
    let mut out = [0u8; 64];
    let rc = unsafe {
        sys::cg_msm_g1(b.as_ptr(), sys::CG_FORM_MONTGOMERY, bases.len() as u64, sc.as_ptr(), scalars.len() as u64, 0, out.as_mut_ptr())
    };
    if rc != 0 {
        return Err(map_err(rc));
    }
    if out.iter().all(|v| *v == 0) {
        return Ok(G1Affine::identity());
    }
    let x = Fq::from_le_bytes_mod_order(&out[..32]);
    let y = Fq::from_le_bytes_mod_order(&out[32..]);
    Ok(G1Affine::new_unchecked(x, y))
}

/// `<G2 as VariableBaseMSM>::msm_bigint` for the show-step sized sums over G2 (forks/ark-poly-commit/src/kzg10/mod.rs:196-290
/// commits in G1; the verifier-side G2 sums of creds/src/utils.rs:124-138 callers are this shape): one-shot.
pub fn msm_g2(bases: &[G2Affine], scalars: &[Fr]) -> Result<G2Affine, SynthesisError> {
    let b = pack_g2(bases);
    let mut sc = Vec::with_capacity(scalars.len() * 32);
    for x in scalars {
        sc.extend_from_slice(&x.into_bigint().to_bytes_le());
    }
    let mut out = [0u8; 128];
    let rc = unsafe {
        sys::cg_msm_g2(b.as_ptr(), sys::CG_FORM_MONTGOMERY, bases.len() as u64, sc.as_ptr(), scalars.len() as u64, 0, out.as_mut_ptr())
    };
    if rc != 0 {
        return Err(map_err(rc));
    }
    if out.iter().all(|v| *v == 0) {
        return Ok(G2Affine::identity());
    }
    let x = Fq2::new(Fq::from_le_bytes_mod_order(&out[..32]), Fq::from_le_bytes_mod_order(&out[32..64]));
    let y = Fq2::new(Fq::from_le_bytes_mod_order(&out[64..96]), Fq::from_le_bytes_mod_order(&out[96..]));
    Ok(G2Affine::new_unchecked(x, y))
}
