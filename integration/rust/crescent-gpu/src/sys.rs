//! Raw bindings: one item per declaration of include/crescent_gpu.h that the shim uses.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const CG_FORM_CANONICAL: u32 = 0;
pub const CG_FORM_MONTGOMERY: u32 = 1;
pub const CG_FLAG_H_COEFFICIENT_BASIS: i32 = 1;
pub const CG_FLAG_LATENCY_MODE: i32 = 2;
pub const CG_FLAG_THROUGHPUT_MODE: i32 = 4;
pub const CG_FLAG_SPIN_WAIT: i32 = 8;
pub const CG_FLAG_CONTIGUOUS_H_SHARDS: i32 = 16;
pub const CG_FLAG_H_SCALARS_EXTERNAL: i32 = 32;
pub const CG_FLAG_STAGED_LOAD: i32 = 64;
pub const CG_FLAG_NO_LONE_SLOT: i32 = 128;
pub const CG_ERR_POLY_DEGREE_TOO_LARGE: c_int = -5;
pub const CG_ERR_MALFORMED_KEY: c_int = -6;

#[repr(C)]
pub struct cg_proving_key {
    pub coord_form: u32,
    pub alpha_g1: *const u8,
    pub beta_g1: *const u8,
    pub delta_g1: *const u8,
    pub beta_g2: *const u8,
    pub delta_g2: *const u8,
    pub a_query: *const u8,
    pub a_len: u64,
    pub b_g1_query: *const u8,
    pub b_g1_len: u64,
    pub b_g2_query: *const u8,
    pub b_g2_len: u64,
    pub h_query: *const u8,
    pub h_len: u64,
    pub l_query: *const u8,
    pub l_len: u64,
}

#[repr(C)]
pub struct cg_csr {
    pub row_ptr: *const u64,
    pub col: *const u32,
    pub coeff: *const u8,
    pub nnz: u64,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct cg_options {
    pub device: i32,
    pub window_bits: i32,
    pub shard_rank: i32,
    pub shard_count: i32,
    pub proof_slots: i32,
    pub flags: i32,
    pub hw_queues: i32,
    pub shard_span: i32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct cg_timings {
    pub upload_ms: f32,
    pub witness_map_ms: f32,
    pub msm_h_ms: f32,
    pub msm_l_ms: f32,
    pub msm_a_ms: f32,
    pub msm_b1_ms: f32,
    pub msm_b2_ms: f32,
    pub finish_ms: f32,
    pub total_ms: f32,
    pub msm_g1_pairs: u64,
    pub msm_g2_pairs: u64,
    pub accum_g1_ms: f32,
    pub accum_g2_ms: f32,
    pub sort_ms: f32,
    pub reserved_ms: f32,
    pub entries_g1: u64,
    pub entries_g2: u64,
    pub accum_g1_launches: u32,
    pub accum_g2_launches: u32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct cg_ctx_info {
    pub table_bytes: u64,
    pub matrix_bytes: u64,
    pub slot_bytes: u64,
    pub total_bytes: u64,
    pub device_free_bytes: u64,
    pub device_total_bytes: u64,
    pub proof_slots: i32,
    pub window_bits: [i32; 5],
    pub tuned: i32,
    pub retune_skipped_for_memory: i32,
    pub retune_attempts: i32,
    pub shard_rank: i32,
    pub shard_count: i32,
    pub latency_mode: i32,
    pub warmup: i32,
    pub lone_slots: i32,
    pub reserved: [i32; 2],
    pub slot_entry_bytes: u64,
    pub slot_piece_bytes: u64,
    pub slot_bucket_bytes: u64,
    pub slot_transform_bytes: u64,
    pub slot_upload_bytes: u64,
    pub lone_slot_bytes: u64,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct cg_load_timings {
    pub total_ms: f32,
    pub matrices_ms: f32,
    pub domain_ms: f32,
    pub key_copy_ms: f32,
    pub fold_ms: f32,
    pub window_tables_ms: f32,
    pub slots_ms: f32,
    pub final_slots_ms: f32,
    pub background_ms: f32,
    pub swap_wait_ms: f32,
    pub ready_after_ms: f32,
    pub staged: i32,
    pub ready: i32,
    pub windows_from_proof: i32,
    pub warmup_proofs: i32,
    pub background_status: i32,
    pub reserved: [i32; 3],
}

pub enum cg_ctx {}
pub enum cg_partial {}
pub enum cg_msm_ctx {}
pub enum cg_qap_ctx {}

extern "C" {
    pub fn cg_init(n_devices: c_int, device_ids: *const c_int) -> c_int;
    pub fn cg_last_error() -> *const c_char;
    pub fn cg_circuit_load(
        out: *mut *mut cg_ctx,
        pk: *const cg_proving_key,
        abc: *const cg_csr,
        num_inputs: u64,
        num_constraints: u64,
        num_variables: u64,
        opt: *const cg_options,
    ) -> c_int;
    pub fn cg_circuit_free(ctx: *mut cg_ctx);
    pub fn cg_prove(
        ctx: *mut cg_ctx,
        full_assignment: *const u8,
        r: *const u8,
        s: *const u8,
        proof_out: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_host_alloc(bytes: u64) -> *mut c_void;
    pub fn cg_host_free(p: *mut c_void);
    pub fn cg_ctx_get_info(ctx: *mut cg_ctx, out: *mut cg_ctx_info) -> c_int;
    pub fn cg_ctx_get_load_timings(ctx: *mut cg_ctx, out: *mut cg_load_timings) -> c_int;
    pub fn cg_ctx_wait_ready(ctx: *mut cg_ctx, timeout_ms: i32) -> c_int;
    // one proof over several GPUs: this shard's five partial sums; the gathered partials finished into a proof
    pub fn cg_prove_partial(
        ctx: *mut cg_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        r: *const u8,
        out_partials: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_assemble(
        ctx: *mut cg_ctx,
        partials: *const u8,
        n_shards: u32,
        r: *const u8,
        s: *const u8,
        proof_out: *mut u8,
    ) -> c_int;
    // the other arrangement: the witness map once, its values scattered, every shard proves with its slice
    pub fn cg_witness_map_coset(
        ctx: *mut cg_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        q_out: *mut c_void,
        q_on_device: c_int,
    ) -> c_int;
    pub fn cg_h_scalars_slice(ctx: *const cg_ctx, shard: u32, offset: *mut u64, count: *mut u64) -> c_int;
    pub fn cg_prove_partial_q(
        ctx: *mut cg_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        q_slice: *const c_void,
        q_on_device: c_int,
        r: *const u8,
        out_partials: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_witness_map(ctx: *mut cg_ctx, full_assignment: *const u8, h_out: *mut u8) -> c_int;
    // the same in two calls: the l, a, b1, b2 sums start at once, the h share follows the slice
    pub fn cg_prove_partial_q_begin(
        ctx: *mut cg_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        r: *const u8,
        out: *mut *mut cg_partial,
    ) -> c_int;
    pub fn cg_partial_witness_map_coset(p: *mut cg_partial, q_out: *mut c_void, q_on_device: c_int) -> c_int;
    pub fn cg_prove_partial_q_finish(
        p: *mut cg_partial,
        q_slice: *const c_void,
        q_on_device: c_int,
        out_partials: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_prove_partial_q_abort(p: *mut cg_partial);
    // the witness map in two halves: which = 0 the a side, 1 the b side; finish2 multiplies a shard's two slices
    pub fn cg_witness_map_coset_half(
        ctx: *mut cg_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        which: c_int,
        out: *mut c_void,
        out_on_device: c_int,
    ) -> c_int;
    pub fn cg_partial_witness_map_coset_half(p: *mut cg_partial, which: c_int, out: *mut c_void, out_on_device: c_int) -> c_int;
    pub fn cg_prove_partial_q_finish2(
        p: *mut cg_partial,
        a_slice: *const c_void,
        b_slice: *const c_void,
        slices_on_device: c_int,
        out_partials: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_domain_size(ctx: *const cg_ctx) -> u64;
    pub fn cg_qap_load(
        out: *mut *mut cg_qap_ctx,
        abc: *const cg_csr,
        num_inputs: u64,
        num_constraints: u64,
        num_variables: u64,
        device: i32,
    ) -> c_int;
    pub fn cg_qap_witness_map(
        ctx: *mut cg_qap_ctx,
        full_assignment: *const c_void,
        assignment_on_device: c_int,
        h_out: *mut c_void,
        h_on_device: c_int,
    ) -> c_int;
    pub fn cg_qap_domain_size(ctx: *const cg_qap_ctx) -> u64;
    pub fn cg_qap_free(ctx: *mut cg_qap_ctx);
    pub fn cg_msm_g1(
        bases: *const u8,
        coord_form: u32,
        n_bases: u64,
        scalars: *const u8,
        n_scalars: u64,
        window_bits: i32,
        out: *mut u8,
    ) -> c_int;
    pub fn cg_msm_g2(
        bases: *const u8,
        coord_form: u32,
        n_bases: u64,
        scalars: *const u8,
        n_scalars: u64,
        window_bits: i32,
        out: *mut u8,
    ) -> c_int;
    pub fn cg_msm_load_g1(
        out: *mut *mut cg_msm_ctx,
        bases: *const u8,
        coord_form: u32,
        n_bases: u64,
        opt: *const cg_options,
    ) -> c_int;
    pub fn cg_msm_run(
        ctx: *mut cg_msm_ctx,
        scalars: *const c_void,
        scalars_on_device: c_int,
        n_scalars: u64,
        out: *mut u8,
        timings: *mut cg_timings,
    ) -> c_int;
    pub fn cg_msm_free(ctx: *mut cg_msm_ctx);
}
