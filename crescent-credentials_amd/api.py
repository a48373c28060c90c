"""ctypes bindings for libcrescent_gpu.so + a host-side mirror of the reference's Groth16 interface.

Names follow `forks/groth16` (Groth16::create_proof_with_reduction_and_matrices, ProvingKey,
VerifyingKey, Proof, LibsnarkReduction::witness_map_from_matrices, generate_parameters_with_qap) so
the parity tests read like the reference's own tests.  All data crossing into the library are packed
little-endian byte arrays (numpy uint8), exactly the C ABI of include/crescent_gpu.h.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRESCENT_GPU_LIB (read by this Python harness, not by the library): load another build of the same ABI - the tuning
# build `libcrescent_gpu_tuning.so` of tools/ab_*.sh and of the fault-injection test
_LIB_PATH = os.environ.get("CRESCENT_GPU_LIB") or os.path.join(_HERE, "libcrescent_gpu.so")
TUNING_LIB_PATH = os.path.join(_HERE, "libcrescent_gpu_tuning.so")

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
CG_FORM_CANONICAL, CG_FORM_MONTGOMERY = 0, 1
CG_FLAG_H_COEFFICIENT_BASIS = 1
CG_FLAG_LATENCY_MODE, CG_FLAG_THROUGHPUT_MODE, CG_FLAG_SPIN_WAIT, CG_FLAG_CONTIGUOUS_H_SHARDS, CG_FLAG_H_SCALARS_EXTERNAL = 2, 4, 8, 16, 32
CG_FLAG_STAGED_LOAD = 64
CG_FLAG_NO_LONE_SLOT = 128


class CrescentGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libcrescent_gpu error %d: %s" % (code, msg))
        self.code = code


def library_path() -> str:
    return _LIB_PATH


class _CgProvingKey(C.Structure):
    _fields_ = [
        ("coord_form", C.c_uint32),
        ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p),
        ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p),
        ("a_query", C.c_void_p), ("a_len", C.c_uint64),
        ("b_g1_query", C.c_void_p), ("b_g1_len", C.c_uint64),
        ("b_g2_query", C.c_void_p), ("b_g2_len", C.c_uint64),
        ("h_query", C.c_void_p), ("h_len", C.c_uint64),
        ("l_query", C.c_void_p), ("l_len", C.c_uint64),
    ]


class _CgCsr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("coeff", C.c_void_p), ("nnz", C.c_uint64)]


class _CgOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("window_bits", C.c_int32), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32), ("proof_slots", C.c_int32), ("flags", C.c_int32), ("hw_queues", C.c_int32), ("shard_span", C.c_int32)]


class CgTimings(C.Structure):
    _fields_ = [("upload_ms", C.c_float), ("witness_map_ms", C.c_float), ("msm_h_ms", C.c_float),
                ("msm_l_ms", C.c_float), ("msm_a_ms", C.c_float), ("msm_b1_ms", C.c_float),
                ("msm_b2_ms", C.c_float), ("finish_ms", C.c_float), ("total_ms", C.c_float),
                ("msm_g1_pairs", C.c_uint64), ("msm_g2_pairs", C.c_uint64),
                ("accum_g1_ms", C.c_float), ("accum_g2_ms", C.c_float), ("sort_ms", C.c_float), ("reserved_ms", C.c_float),
                ("entries_g1", C.c_uint64), ("entries_g2", C.c_uint64),
                ("accum_g1_launches", C.c_uint32), ("accum_g2_launches", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CgCtxInfo(C.Structure):
    _fields_ = [("table_bytes", C.c_uint64), ("matrix_bytes", C.c_uint64), ("slot_bytes", C.c_uint64), ("total_bytes", C.c_uint64),
                ("device_free_bytes", C.c_uint64), ("device_total_bytes", C.c_uint64), ("proof_slots", C.c_int32),
                ("window_bits", C.c_int32 * 5), ("tuned", C.c_int32), ("retune_skipped_for_memory", C.c_int32),
                ("retune_attempts", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("latency_mode", C.c_int32),
                ("warmup", C.c_int32), ("lone_slots", C.c_int32), ("reserved", C.c_int32 * 2), ("slot_entry_bytes", C.c_uint64),
                ("slot_piece_bytes", C.c_uint64), ("slot_bucket_bytes", C.c_uint64), ("slot_transform_bytes", C.c_uint64),
                ("slot_upload_bytes", C.c_uint64), ("lone_slot_bytes", C.c_uint64)]


class CgLoadTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("total_ms", "matrices_ms", "domain_ms", "key_copy_ms", "fold_ms", "window_tables_ms", "slots_ms",
                                         "final_slots_ms", "background_ms", "swap_wait_ms", "ready_after_ms")] + \
               [(n, C.c_int32) for n in ("staged", "ready", "windows_from_proof", "warmup_proofs", "background_status")] + \
               [("reserved", C.c_int32 * 3)]


class _CgProverParamsView(C.Structure):
    _fields_ = [("pk", _CgProvingKey), ("gamma_g2", C.c_void_p), ("gamma_abc_g1", C.c_void_p), ("gamma_abc_len", C.c_uint64),
                ("vk_bytes", C.c_void_p), ("vk_len", C.c_uint64), ("pvk_bytes", C.c_void_p), ("pvk_len", C.c_uint64),
                ("config_str", C.c_void_p), ("config_len", C.c_uint64)]


class _CgClientStateView(C.Structure):
    _fields_ = [("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("aux", C.c_void_p), ("aux_len", C.c_uint64),
                ("has_aux", C.c_int32), ("has_input_com_randomness", C.c_int32), ("proof", C.c_void_p),
                ("vk_bytes", C.c_void_p), ("vk_len", C.c_uint64), ("pvk_bytes", C.c_void_p), ("pvk_len", C.c_uint64),
                ("input_com_randomness", C.c_void_p),
                ("openings_bytes", C.c_void_p), ("openings_len", C.c_uint64), ("n_openings", C.c_uint64),
                ("credtype", C.c_void_p), ("credtype_len", C.c_uint64), ("config_str", C.c_void_p), ("config_len", C.c_uint64)]


class _CgR1csHeader(C.Structure):
    _fields_ = [("field_size", C.c_uint32), ("n_wires", C.c_uint32), ("n_pub_out", C.c_uint32),
                ("n_pub_in", C.c_uint32), ("n_prv_in", C.c_uint32), ("n_constraints", C.c_uint32),
                ("n_labels", C.c_uint64), ("num_inputs", C.c_uint64), ("num_variables", C.c_uint64)]


# every symbol include/crescent_gpu.h declares, with its signature
_SIGNATURES = {
    "cg_init": (C.c_int, [C.c_int, C.c_void_p]),
    "cg_set_device": (C.c_int, [C.c_int32]),
    "cg_last_error": (C.c_char_p, []),
    "cg_version": (C.c_char_p, []),
    "cg_circuit_load": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(_CgProvingKey), C.POINTER(_CgCsr), C.c_uint64,
                                  C.c_uint64, C.c_uint64, C.POINTER(_CgOptions)]),
    "cg_circuit_free": (None, [C.c_void_p]),
    "cg_prove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_prove_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_prove_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_witness_map_coset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cg_h_scalars_slice": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cg_prove_partial_q": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_prove_partial_q_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cg_partial_witness_map_coset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "cg_prove_partial_q_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_prove_partial_q_abort": (None, [C.c_void_p]),
    "cg_witness_map_coset_half": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "cg_partial_witness_map_coset_half": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cg_prove_partial_q_finish2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_host_alloc": (C.c_void_p, [C.c_uint64]),
    "cg_host_free": (None, [C.c_void_p]),
    "cg_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cg_host_unregister": (C.c_int, [C.c_void_p]),
    "cg_ctx_get_info": (C.c_int, [C.c_void_p, C.POINTER(CgCtxInfo)]),
    "cg_ctx_get_load_timings": (C.c_int, [C.c_void_p, C.POINTER(CgLoadTimings)]),
    "cg_ctx_wait_ready": (C.c_int, [C.c_void_p, C.c_int32]),
    "cg_probe_shader_clock": (C.c_int, [C.c_int32, C.c_uint32, C.POINTER(C.c_double)]),
    "cg_witness_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_domain_size": (C.c_uint64, [C.c_void_p]),
    "cg_qap_load": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(_CgCsr), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32]),
    "cg_qap_witness_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cg_qap_domain_size": (C.c_uint64, [C.c_void_p]),
    "cg_qap_free": (None, [C.c_void_p]),
    "cg_msm_g1": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]),
    "cg_msm_g2": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]),
    "cg_ntt": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "cg_msm_load_g1": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(_CgOptions)]),
    "cg_msm_load_g2": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(_CgOptions)]),
    "cg_msm_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(CgTimings)]),
    "cg_msm_free": (None, [C.c_void_p]),
    "cg_ntt_load": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_int32]),
    "cg_ntt_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "cg_ntt_free": (None, [C.c_void_p]),
    "cg_fixed_base_g1": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "cg_fixed_base_g2": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "cg_setup": (C.c_int, [C.POINTER(_CgCsr), C.c_uint64, C.c_uint64, C.c_uint64] + [C.c_void_p] * 11),
    "cg_r1cs_parse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cg_r1cs_get": (C.c_int, [C.c_void_p, C.POINTER(_CgR1csHeader), C.POINTER(_CgCsr), C.POINTER(C.c_void_p)]),
    "cg_r1cs_free": (None, [C.c_void_p]),
    "cg_pk_parse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "cg_pk_get": (C.c_int, [C.c_void_p, C.POINTER(_CgProvingKey), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "cg_pk_free": (None, [C.c_void_p]),
    "cg_pk_serialized_size": (C.c_uint64, [C.POINTER(_CgProvingKey), C.c_uint64]),
    "cg_pk_serialize": (C.c_int, [C.POINTER(_CgProvingKey), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "cg_prover_params_parse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cg_prover_params_get": (C.c_int, [C.c_void_p, C.POINTER(_CgProverParamsView)]),
    "cg_prover_params_free": (None, [C.c_void_p]),
    "cg_prover_params_serialized_size": (C.c_uint64, [C.POINTER(_CgProvingKey), C.c_uint64, C.c_uint64, C.c_uint64]),
    "cg_prover_params_serialize": (C.c_int, [C.POINTER(_CgProvingKey), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "cg_client_state_serialized_size": (C.c_uint64, [C.POINTER(_CgClientStateView)]),
    "cg_client_state_serialize": (C.c_int, [C.POINTER(_CgClientStateView), C.c_void_p, C.c_uint64]),
    "cg_client_state_parse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cg_client_state_get": (C.c_int, [C.c_void_p, C.POINTER(_CgClientStateView)]),
    "cg_client_state_free": (None, [C.c_void_p]),
}

_lib = None


def _init_torch_runtime_first() -> None:
    """A Python process that uses torch tensors next to this library holds TWO HIP runtimes: the ROCm one
    libcrescent_gpu.so links and the copy bundled in the torch wheel (different SONAMEs, so the loader keeps both).
    Measured on the MI355X pool: with the bundled runtime initialised first both work; the other way round torch then
    reports "No HIP GPUs are available".  So when torch is present its runtime is brought up before the library is
    loaded.  Hosts without torch (the Rust shim) are not affected."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def lib() -> C.CDLL:
    """The HIP library.  Raises if it has not been built: there is deliberately no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise CrescentGpuError(-2, "%s not found; run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB_PATH)
        _init_torch_runtime_first()
        L = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise CrescentGpuError(rc, lib().cg_last_error().decode("utf-8", "replace"))


def _u8(a, nbytes: Optional[int] = None) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    if nbytes is not None and a.size != nbytes:
        raise ValueError("expected %d bytes, got %d" % (nbytes, a.size))
    return a


def fr_to_bytes(x: int) -> bytes:
    return int(x % FR_MODULUS).to_bytes(32, "little")


def scalars_to_array(vals: Sequence[int]) -> np.ndarray:
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


# ------------------------------------------------------------------------------------------------
# data structures (forks/groth16/src/data_structures.rs)
# ------------------------------------------------------------------------------------------------
@dataclass
class VerifyingKey:
    """data_structures.rs:31-44 (fork adds delta_g1).  Packed canonical bytes."""
    alpha_g1: np.ndarray
    beta_g2: np.ndarray
    gamma_g2: np.ndarray
    delta_g1: np.ndarray
    delta_g2: np.ndarray
    gamma_abc_g1: np.ndarray      # num_inputs x 64


@dataclass
class ProvingKey:
    """data_structures.rs:101-118.  Queries are packed arrays: G1 64 B, G2 128 B per point, identity = zeros."""
    vk: VerifyingKey
    beta_g1: np.ndarray
    delta_g1: np.ndarray
    a_query: np.ndarray
    b_g1_query: np.ndarray
    b_g2_query: np.ndarray
    h_query: np.ndarray
    l_query: np.ndarray
    coord_form: int = CG_FORM_CANONICAL

    def _c(self) -> _CgProvingKey:
        k = _CgProvingKey()
        k.coord_form = self.coord_form
        k.alpha_g1 = _ptr(self.vk.alpha_g1); k.beta_g1 = _ptr(self.beta_g1); k.delta_g1 = _ptr(self.delta_g1)
        k.beta_g2 = _ptr(self.vk.beta_g2); k.delta_g2 = _ptr(self.vk.delta_g2)
        k.a_query = _ptr(self.a_query); k.a_len = self.a_query.size // 64
        k.b_g1_query = _ptr(self.b_g1_query); k.b_g1_len = self.b_g1_query.size // 64
        k.b_g2_query = _ptr(self.b_g2_query); k.b_g2_len = self.b_g2_query.size // 128
        k.h_query = _ptr(self.h_query); k.h_len = self.h_query.size // 64
        k.l_query = _ptr(self.l_query); k.l_len = self.l_query.size // 64
        return k


@dataclass
class Proof:
    """data_structures.rs:7-14; `data` is the 256-byte ark-serialize uncompressed a ‖ b ‖ c."""
    data: bytes

    def serialize_uncompressed(self) -> bytes:
        return self.data

    @property
    def a(self) -> bytes:
        return self.data[:64]

    @property
    def b(self) -> bytes:
        return self.data[64:192]

    @property
    def c(self) -> bytes:
        return self.data[192:]


@dataclass
class _Csr:
    row_ptr: np.ndarray   # uint64, rows + 1
    col: np.ndarray       # uint32
    coeff: np.ndarray     # uint8, nnz * 32 canonical

    @property
    def nnz(self) -> int:
        return int(self.col.size)


class ConstraintMatrices:
    """ark-relations ConstraintMatrices<F> (fields visible at forks/circom-compat/src/zkey.rs:181-193),
    held as three CSR matrices."""

    def __init__(self, a: _Csr, b: _Csr, c: _Csr, num_instance_variables: int, num_witness_variables: int,
                 num_constraints: int, _keepalive=None):
        self.a, self.b, self.c = a, b, c
        self.num_instance_variables = num_instance_variables
        self.num_witness_variables = num_witness_variables
        self.num_constraints = num_constraints
        self._keepalive = _keepalive

    @staticmethod
    def from_rows(A, B, Cm, num_instance_variables: int, num_variables: int) -> "ConstraintMatrices":
        """rows: list of list of (coeff:int, column:int) — the Vec<Vec<(F, usize)>> form."""
        def conv(rows):
            rp = np.zeros(len(rows) + 1, dtype=np.uint64)
            cols, coefs = [], []
            for i, row in enumerate(rows):
                for coeff, col in row:
                    cols.append(col)
                    coefs.append(int(coeff % FR_MODULUS).to_bytes(32, "little"))
                rp[i + 1] = len(cols)
            return _Csr(rp, np.asarray(cols, dtype=np.uint32), np.frombuffer(b"".join(coefs), dtype=np.uint8).copy()
                        if coefs else np.zeros(0, dtype=np.uint8))
        return ConstraintMatrices(conv(A), conv(B), conv(Cm), num_instance_variables,
                                  num_variables - num_instance_variables, len(A))

    @property
    def num_variables(self) -> int:
        return self.num_instance_variables + self.num_witness_variables

    def _c(self):
        arr = (_CgCsr * 3)()
        keep = []
        for i, m in enumerate((self.a, self.b, self.c)):
            col = m.col if m.col.size else np.zeros(1, dtype=np.uint32)
            coeff = m.coeff if m.coeff.size else np.zeros(32, dtype=np.uint8)
            keep += [col, coeff]
            arr[i].row_ptr = _ptr(m.row_ptr); arr[i].col = _ptr(col); arr[i].coeff = _ptr(coeff); arr[i].nnz = m.nnz
        return arr, keep


# ------------------------------------------------------------------------------------------------
# prover
# ------------------------------------------------------------------------------------------------
class OpenPartial:
    """a sharded proof between cg_prove_partial_q_begin and _finish (cg_partial): it holds one of its context's proof slots"""

    def __init__(self, prover, handle, keep):
        self._prover, self._h, self._keep = prover, handle, keep

    def witness_map_coset(self, out_dev: Optional[int] = None, out_host: Optional[int] = None):
        """cg_partial_witness_map_coset (the rank that runs the witness map): all coset values for this proof's assignment,
        shard-major; -> numpy bytes, or written to the device address out_dev / the host address out_host"""
        if out_dev is not None:
            _check(lib().cg_partial_witness_map_coset(self._h, C.c_void_p(int(out_dev)), 1))
            return None
        if out_host is not None:
            _check(lib().cg_partial_witness_map_coset(self._h, C.c_void_p(int(out_host)), 0))
            return None
        q = np.zeros(self._prover.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_partial_witness_map_coset(self._h, _ptr(q), 0))
        return q

    def witness_map_coset_half(self, which: int, out_dev: Optional[int] = None, out_host: Optional[int] = None):
        """cg_partial_witness_map_coset_half: ONE side of the coset values (which = 0: vinv·a, 1: b) for this proof's assignment"""
        if out_dev is not None:
            _check(lib().cg_partial_witness_map_coset_half(self._h, which, C.c_void_p(int(out_dev)), 1))
            return None
        if out_host is not None:
            _check(lib().cg_partial_witness_map_coset_half(self._h, which, C.c_void_p(int(out_host)), 0))
            return None
        q = np.zeros(self._prover.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_partial_witness_map_coset_half(self._h, which, _ptr(q), 0))
        return q

    def finish2(self, a_slice, b_slice, on_device: bool = False, timings: bool = False):
        """cg_prove_partial_q_finish2: the h share from this shard's slices of BOTH sides (their products are the h scalars)"""
        out = np.zeros(384, dtype=np.uint8)
        if on_device:
            ap, bp = C.c_void_p(int(a_slice)), C.c_void_p(int(b_slice))
        else:
            a, b = _u8(a_slice), _u8(b_slice)
            ap, bp = C.c_void_p(_ptr(a) if a.size else _ptr(out)), C.c_void_p(_ptr(b) if b.size else _ptr(out))
        tm = CgTimings()
        h, self._h = self._h, None
        _check(lib().cg_prove_partial_q_finish2(h, ap, bp, 1 if on_device else 0, _ptr(out), C.byref(tm) if timings else None))
        return (out.tobytes(), tm.as_dict()) if timings else out.tobytes()

    def finish(self, q_slice, q_on_device: bool = False, timings: bool = False):
        """cg_prove_partial_q_finish: the h share with this shard's slice; -> the 384-byte record.  The handle is gone afterwards,
        whether the call succeeded or not."""
        out = np.zeros(384, dtype=np.uint8)
        if q_on_device:
            qptr = C.c_void_p(int(q_slice))
        else:
            q = _u8(q_slice)
            qptr = C.c_void_p(_ptr(q) if q.size else _ptr(out))
        tm = CgTimings()
        h, self._h = self._h, None
        _check(lib().cg_prove_partial_q_finish(h, qptr, 1 if q_on_device else 0, _ptr(out), C.byref(tm) if timings else None))
        return (out.tobytes(), tm.as_dict()) if timings else out.tobytes()

    def abort(self):
        if self._h is not None:
            h, self._h = self._h, None
            lib().cg_prove_partial_q_abort(h)

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class Prover:
    """A circuit loaded on one GPU (cg_ctx): proving key tables + matrices resident in HBM."""

    def __init__(self, pk: ProvingKey, matrices: ConstraintMatrices, device: int = -1, window_bits: int = 0,
                 shard_rank: int = 0, shard_count: int = 1, proof_slots: int = 1, h_coefficient_basis: bool = False,
                 mode: Optional[str] = None, spin_wait: bool = False, contiguous_h_shards: bool = False,
                 h_scalars_external: bool = False, flags: int = 0, staged_load: bool = False, shard_span: Optional[Tuple[int, int]] = None,
                 lone_slot: bool = True):
        """lone_slot=False: CG_FLAG_NO_LONE_SLOT (a throughput context then holds no extra slot for proofs that arrive alone).
        h_coefficient_basis=True keeps the h query as loaded (seven transforms per proof, CG_FLAG_H_COEFFICIENT_BASIS).
        shard_span: (lo, hi) in 1/10000 of every query - this shard's part instead of the shard_rank-th of shard_count equal parts
        (cg_options.shard_span: unequal shares for the ranks that also run the witness map).
        staged_load: CG_FLAG_STAGED_LOAD - the call returns as soon as the context can prove (warm-up arrangement) and a worker
        thread of the library finishes the load behind the first proofs (wait_ready(), load_timings()).
        mode: None (proof_slots decides), "latency" or "throughput" (CG_FLAG_LATENCY_MODE / CG_FLAG_THROUGHPUT_MODE);
        spin_wait: CG_FLAG_SPIN_WAIT; contiguous_h_shards: CG_FLAG_CONTIGUOUS_H_SHARDS; h_scalars_external:
        CG_FLAG_H_SCALARS_EXTERNAL (a shard that never runs the witness map: prove_partial_q only); flags: further raw bits."""
        if mode not in (None, "latency", "throughput"):
            raise ValueError("mode must be None, 'latency' or 'throughput'")
        flags |= (CG_FLAG_H_COEFFICIENT_BASIS if h_coefficient_basis else 0) | (CG_FLAG_LATENCY_MODE if mode == "latency" else 0) | \
                 (CG_FLAG_THROUGHPUT_MODE if mode == "throughput" else 0) | (CG_FLAG_SPIN_WAIT if spin_wait else 0) | \
                 (CG_FLAG_CONTIGUOUS_H_SHARDS if contiguous_h_shards else 0) | (CG_FLAG_H_SCALARS_EXTERNAL if h_scalars_external else 0) | \
                 (CG_FLAG_STAGED_LOAD if staged_load else 0) | (0 if lone_slot else CG_FLAG_NO_LONE_SLOT)
        L = lib()
        self.num_inputs = matrices.num_instance_variables
        self.num_constraints = matrices.num_constraints
        self.num_variables = matrices.num_variables
        opt = _CgOptions(device=device, window_bits=window_bits, shard_rank=shard_rank, shard_count=shard_count,
                         proof_slots=proof_slots, flags=flags, shard_span=(shard_span[0] | shard_span[1] << 16) if shard_span else 0)
        self.proof_slots = max(1, proof_slots)
        cpk = pk._c()
        abc, _keep = matrices._c()
        h = C.c_void_p()
        _check(L.cg_circuit_load(C.byref(h), C.byref(cpk), abc, self.num_inputs, self.num_constraints,
                                 self.num_variables, C.byref(opt)))
        self._h = h
        self.domain_size = int(L.cg_domain_size(h))
        self._lease_lock = threading.Lock()
        self._leases = 0                 # callers inside a cached prover (Groth16._prover_for); close() is deferred while > 0
        self._close_when_idle = False

    def close(self):
        if getattr(self, "_h", None):
            lib().cg_circuit_free(self._h)
            self._h = None

    def _lease(self) -> "Prover":
        with self._lease_lock:
            self._leases += 1
        return self

    def _unlease(self) -> None:
        with self._lease_lock:
            self._leases -= 1
            retire = self._leases == 0 and self._close_when_idle
        if retire:
            self.close()

    def _retire(self) -> None:
        """evicted from a cache: closed now if nobody is inside it, otherwise by the last caller to leave"""
        with self._lease_lock:
            self._close_when_idle = True
            idle = self._leases == 0
        if idle:
            self.close()

    def info(self) -> dict:
        """cg_ctx_get_info: resident bytes (tables, matrices, per slot), the current window of each MSM, whether the
        one-time re-tune happened or was skipped for lack of memory"""
        ci = CgCtxInfo()
        _check(lib().cg_ctx_get_info(self._h, C.byref(ci)))
        d = {k: getattr(ci, k) for k, _ in ci._fields_ if k not in ("window_bits", "reserved")}
        d["window_bits"] = dict(zip(("h", "l", "a", "b_g1", "b_g2"), list(ci.window_bits)))
        return d

    def load_timings(self) -> dict:
        """cg_ctx_get_load_timings: where the time of cg_circuit_load went; for a staged load also the background part"""
        lt = CgLoadTimings()
        _check(lib().cg_ctx_get_load_timings(self._h, C.byref(lt)))
        return {k: (round(getattr(lt, k), 3) if isinstance(getattr(lt, k), float) else getattr(lt, k)) for k, _ in lt._fields_ if k != "reserved"}

    def wait_ready(self, timeout_ms: int = -1) -> bool:
        """cg_ctx_wait_ready: True once the final arrangement is in force, False when the time ran out first; raises when the
        background part of a staged load failed"""
        rc = lib().cg_ctx_wait_ready(self._h, timeout_ms)
        if rc == 1:
            return False
        _check(rc)
        return True

    def prove_host_ptr(self, ptr: int, r: int, s: int, timings: bool = False):
        """cg_prove on a raw host address (num_variables x 32 B canonical): pageable or page-locked (HostBuffer)"""
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(fr_to_bytes(r)), _u8(fr_to_bytes(s))
        tm = CgTimings()
        _check(lib().cg_prove(self._h, C.c_void_p(ptr), _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm) if timings else None))
        p = Proof(out.tobytes())
        return (p, tm.as_dict()) if timings else p

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove(self, full_assignment, r: int, s: int, timings: bool = False):
        w = _u8(full_assignment, self.num_variables * 32)
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(fr_to_bytes(r)), _u8(fr_to_bytes(s))
        tm = CgTimings()
        _check(lib().cg_prove(self._h, _ptr(w), _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm) if timings else None))
        p = Proof(out.tobytes())
        return (p, tm.as_dict()) if timings else p

    def prove_dev(self, d_ptr: int, r: int, s: int, timings: bool = False):
        """assignment already in this GPU's memory (e.g. a torch uint8 tensor's data_ptr())."""
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(fr_to_bytes(r)), _u8(fr_to_bytes(s))
        tm = CgTimings()
        _check(lib().cg_prove_dev(self._h, C.c_void_p(d_ptr), _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm) if timings else None))
        p = Proof(out.tobytes())
        return (p, tm.as_dict()) if timings else p

    def prove_partial(self, assignment, r: int, on_device: bool = False, timings: bool = False):
        out = np.zeros(384, dtype=np.uint8)
        rb = _u8(fr_to_bytes(r))
        if on_device:
            ptr = C.c_void_p(int(assignment))
        else:
            w = _u8(assignment, self.num_variables * 32)
            ptr = C.c_void_p(_ptr(w))
        tm = CgTimings()
        _check(lib().cg_prove_partial(self._h, ptr, 1 if on_device else 0, _ptr(rb), _ptr(out), C.byref(tm) if timings else None))
        return (out.tobytes(), tm.as_dict()) if timings else out.tobytes()

    def prove_partial_q(self, assignment, q_slice, r: int, on_device: bool = False, q_on_device: bool = False, timings: bool = False):
        """cg_prove_partial_q: this shard's partial sums with its h scalars supplied (its slice of witness_map_coset's
        output: h_scalars_slice); the witness map is skipped.  assignment / q_slice: numpy bytes, or device addresses."""
        out = np.zeros(384, dtype=np.uint8)
        rb = _u8(fr_to_bytes(r))
        if on_device:
            ptr = C.c_void_p(int(assignment))
        else:
            w = _u8(assignment, self.num_variables * 32)
            ptr = C.c_void_p(_ptr(w))
        if q_on_device:
            qptr = C.c_void_p(int(q_slice))
        else:
            q = _u8(q_slice)
            qptr = C.c_void_p(_ptr(q) if q.size else _ptr(rb))
        tm = CgTimings()
        _check(lib().cg_prove_partial_q(self._h, ptr, 1 if on_device else 0, qptr, 1 if q_on_device else 0, _ptr(rb), _ptr(out),
                                        C.byref(tm) if timings else None))
        return (out.tobytes(), tm.as_dict()) if timings else out.tobytes()

    def prove_partial_q_begin(self, assignment, r: int, on_device: bool = False) -> "OpenPartial":
        """cg_prove_partial_q_begin: takes a proof slot, queues this shard's l, a, b1, b2 partial sums and returns while they run;
        the h share follows with OpenPartial.finish(slice) once the slice has arrived"""
        rb = _u8(fr_to_bytes(r))
        keep = None
        if on_device:
            ptr = C.c_void_p(int(assignment))
        else:
            keep = _u8(assignment, self.num_variables * 32)
            ptr = C.c_void_p(_ptr(keep))
        h = C.c_void_p()
        _check(lib().cg_prove_partial_q_begin(self._h, ptr, 1 if on_device else 0, _ptr(rb), C.byref(h)))
        return OpenPartial(self, h, keep)

    def h_scalars_slice(self, shard: int) -> Tuple[int, int]:
        """(offset, count) of shard `shard`'s scalars inside witness_map_coset's output, in elements (cg_h_scalars_slice)"""
        off, cnt = C.c_uint64(), C.c_uint64()
        _check(lib().cg_h_scalars_slice(self._h, shard, C.byref(off), C.byref(cnt)))
        return int(off.value), int(cnt.value)

    def witness_map_coset(self, assignment, on_device: bool = False, out_dev: Optional[int] = None, out_host: Optional[int] = None):
        """cg_witness_map_coset: ALL coset values q_j of the quotient's a·b part (folded key), domain_size x 32 B canonical,
        shard-major for this context's shard count.  -> numpy bytes, or written to the device address out_dev / the host
        address out_host (domain_size x 32 writable bytes)."""
        if on_device:
            ptr = C.c_void_p(int(assignment))
        else:
            w = _u8(assignment, self.num_variables * 32)
            ptr = C.c_void_p(_ptr(w))
        if out_dev is not None:
            _check(lib().cg_witness_map_coset(self._h, ptr, 1 if on_device else 0, C.c_void_p(int(out_dev)), 1))
            return None
        if out_host is not None:
            _check(lib().cg_witness_map_coset(self._h, ptr, 1 if on_device else 0, C.c_void_p(int(out_host)), 0))
            return None
        q = np.zeros(self.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_witness_map_coset(self._h, ptr, 1 if on_device else 0, _ptr(q), 0))
        return q

    def witness_map_coset_half(self, assignment, which: int, on_device: bool = False, out_dev: Optional[int] = None,
                               out_host: Optional[int] = None):
        """cg_witness_map_coset_half: ONE side of the coset values - which = 0: vinv·a(g w^j), 1: b(g w^j) - as plain canonical
        integers, laid out like witness_map_coset's output; q is their product mod r"""
        if on_device:
            ptr = C.c_void_p(int(assignment))
        else:
            w = _u8(assignment, self.num_variables * 32)
            ptr = C.c_void_p(_ptr(w))
        if out_dev is not None:
            _check(lib().cg_witness_map_coset_half(self._h, ptr, 1 if on_device else 0, which, C.c_void_p(int(out_dev)), 1))
            return None
        if out_host is not None:
            _check(lib().cg_witness_map_coset_half(self._h, ptr, 1 if on_device else 0, which, C.c_void_p(int(out_host)), 0))
            return None
        q = np.zeros(self.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_witness_map_coset_half(self._h, ptr, 1 if on_device else 0, which, _ptr(q), 0))
        return q

    def assemble(self, partials: bytes, n_shards: int, r: int, s: int) -> Proof:
        pb = _u8(partials, 384 * n_shards)
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(fr_to_bytes(r)), _u8(fr_to_bytes(s))
        _check(lib().cg_assemble(self._h, _ptr(pb), n_shards, _ptr(rb), _ptr(sb), _ptr(out)))
        return Proof(out.tobytes())

    def witness_map(self, full_assignment) -> np.ndarray:
        w = _u8(full_assignment, self.num_variables * 32)
        h = np.zeros(self.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_witness_map(self._h, _ptr(w), _ptr(h)))
        return h


class HostBuffer:
    """Page-locked host memory from cg_host_alloc, viewed as a numpy uint8 array (`.array`): where a host lets the
    witness calculator write the full assignment so that cg_prove's upload is one asynchronous DMA."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = lib().cg_host_alloc(self.nbytes)
        if not self.ptr:
            raise CrescentGpuError(-4, lib().cg_last_error().decode("utf-8", "replace"))
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.nbytes,))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().cg_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_device(device: int) -> None:
    """cg_set_device: the GPU that the entry points without a device of their own (the setup, the one-shot MSM / NTT) use
    from this thread on.  Needed next to torch: the library links its own HIP runtime, which does not see
    torch.cuda.set_device."""
    _check(lib().cg_set_device(device))


def probe_shader_clock(device: int = -1, window_us: int = 20000) -> float:
    """GHz the shader engines hold over the next `window_us` (cg_probe_shader_clock); safe to call from a second thread
    while proofs run"""
    g = C.c_double(0.0)
    _check(lib().cg_probe_shader_clock(device, window_us, C.byref(g)))
    return float(g.value)


def host_register(a: np.ndarray) -> None:
    """pin an existing contiguous numpy buffer in place (cg_host_register); undo with host_unregister before freeing it"""
    _check(lib().cg_host_register(C.c_void_p(a.ctypes.data), a.nbytes))


def host_unregister(a: np.ndarray) -> None:
    _check(lib().cg_host_unregister(C.c_void_p(a.ctypes.data)))


class QapContext:
    """Three constraint matrices resident on one GPU with their domain tables, no proving key (cg_qap_ctx)."""

    def __init__(self, matrices: ConstraintMatrices, device: int = -1):
        self.num_inputs = matrices.num_instance_variables
        self.num_constraints = matrices.num_constraints
        self.num_variables = matrices.num_variables
        abc, _keep = matrices._c()
        self._h = C.c_void_p()
        _check(lib().cg_qap_load(C.byref(self._h), abc, self.num_inputs, self.num_constraints, self.num_variables, device))
        self.domain_size = int(lib().cg_qap_domain_size(self._h))
        self._lease_lock = threading.Lock()
        self._leases = 0
        self._close_when_idle = False

    _lease = Prover._lease
    _unlease = Prover._unlease
    _retire = Prover._retire

    def witness_map(self, full_assignment) -> np.ndarray:
        w = _u8(full_assignment, self.num_variables * 32)
        h = np.zeros(self.domain_size * 32, dtype=np.uint8)
        _check(lib().cg_qap_witness_map(self._h, _ptr(w), 0, _ptr(h), 0))
        return h

    def witness_map_dev(self, d_assignment: int, d_h: int) -> None:
        """assignment (num_variables x 32 B) and h (domain_size x 32 B) both in this GPU's memory"""
        _check(lib().cg_qap_witness_map(self._h, C.c_void_p(d_assignment), 1, C.c_void_p(d_h), 1))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().cg_qap_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class R1CSToQAP:
    """forks/groth16/src/r1cs_to_qap.rs:49-98: the trait `Groth16<E, QAP>` is generic over (lib.rs:55-57).  The
    prover's half is `witness_map_from_matrices`; the generator's half (`instance_map_with_evaluation`,
    `h_query_scalars`, :52-56,82-97) is what `generate_parameters_with_qap` -> cg_setup evaluates on the GPU."""

    @classmethod
    def witness_map_from_matrices(cls, matrices: ConstraintMatrices, num_inputs: int, num_constraints: int,
                                  full_assignment) -> np.ndarray:
        raise NotImplementedError


def _content_digest(*arrays) -> bytes:
    """128-bit digest of the bytes of the given numpy arrays / bytes objects (xxh3 when present - tens of milliseconds for
    the 0.6 GB of a full-size circuit - else blake2b).  Cache keys only: never part of a proof."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except Exception:
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a)
            h.update(np.int64(a.size).tobytes())
            h.update(a.view(np.uint8).reshape(-1).data if a.size else b"")
        else:
            h.update(np.int64(len(a)).tobytes())
            h.update(a)
    return h.digest()


class _Held:
    """The arrays a content digest was computed over, HELD (so that neither their ids nor their addresses can be recycled by
    an array that replaces them: the memo of rounds 3-4 kept (id, address, size) only, and a dropped array's id and address
    are commonly handed to the next one of the same size) and compared by identity."""

    def __init__(self, arrays):
        self.arrays = tuple(arrays)

    def same_as(self, arrays) -> bool:
        return len(arrays) == len(self.arrays) and all(a is b for a, b in zip(arrays, self.arrays))


def _freeze(*arrays) -> "_Held":
    """The digest of an object is remembered on it, so its arrays must not change afterwards: the arrays that were HASHED are
    made read-only, so that an in-place write through them raises instead of silently proving against the old resident copy.
    Only those: an array they are views of stays as the caller left it (round 5 walked `.base` and froze the parents too, which
    changed unrelated views of a buffer the caller owns).  A view whose parent is still writable can be changed behind the
    memo's back, so such a set is marked not `stable` and its digest is NOT remembered: it is recomputed at every use (0.3 s per
    GB) - slower, never stale.  An attribute REPLACED by another array (the supported way to change a key or a matrix) is seen
    - the memo holds the arrays themselves - and hashed again."""
    stable = True
    for a in arrays:
        if isinstance(a, np.ndarray):
            try:
                a.setflags(write=False)
            except ValueError:
                pass
            b = a.base
            while isinstance(b, np.ndarray):
                if b.flags.writeable:
                    stable = False
                b = b.base
    held = _Held(arrays)
    held.stable = stable
    return held


def _matrices_key(m: "ConstraintMatrices") -> tuple:
    """what the three matrices ARE, not where they live: shape + a digest of every index and coefficient (computed once
    per object and remembered on it; the arrays are immutable from then on, see _freeze)"""
    arrays = (m.a.row_ptr, m.a.col, m.a.coeff, m.b.row_ptr, m.b.col, m.b.coeff, m.c.row_ptr, m.c.col, m.c.coeff)
    memo = getattr(m, "_content_key", None)
    if memo is not None and memo[0].same_as(arrays):
        return memo[1]
    k = (m.num_instance_variables, m.num_witness_variables, m.num_constraints, m.a.nnz, m.b.nnz, m.c.nnz, _content_digest(*arrays))
    held = _freeze(*arrays)
    m._content_key = (held, k) if held.stable else None
    return k


def _pk_key(pk: "ProvingKey") -> tuple:
    """a proving key by content: the verifying key's points, the query lengths and a digest of the queries"""
    arrays = (pk.vk.alpha_g1, pk.vk.beta_g2, pk.vk.gamma_g2, pk.vk.delta_g1, pk.vk.delta_g2, pk.vk.gamma_abc_g1,
              pk.beta_g1, pk.delta_g1, pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query)
    memo = getattr(pk, "_content_key", None)
    if memo is not None and memo[0].same_as(arrays) and memo[2] == pk.coord_form:
        return memo[1]
    k = (pk.coord_form, pk.a_query.size, pk.h_query.size, pk.l_query.size, _content_digest(*arrays))
    held = _freeze(*arrays)
    pk._content_key = (held, k, pk.coord_form) if held.stable else None
    return k


class _ResidentCache:
    """Resident GPU contexts keyed by the CONTENT of what they were loaded from, least recently used first out, safe to
    use from several threads: lookups and evictions happen under one lock, a context handed out is leased (see
    Prover._lease) and an evicted context is closed by the last caller to leave it, never under one."""

    def __init__(self, max_cached: int):
        self.max_cached = max_cached
        self._lock = threading.Lock()
        self._entries = {}          # key -> context; insertion order = recency
        self._loading = {}          # key -> threading.Event: one thread loads, the others wait for it

    def lease(self, key, make):
        while True:
            with self._lock:
                ctx = self._entries.pop(key, None)
                if ctx is not None:
                    self._entries[key] = ctx                  # most recent
                    return ctx._lease()
                ev = self._loading.get(key)
                if ev is None:
                    ev = self._loading[key] = threading.Event()
                    break
            ev.wait()                                         # another thread is loading this very circuit
        try:
            ctx = make()                                      # seconds: outside the lock
        except BaseException:
            with self._lock:
                self._loading.pop(key).set()
            raise
        evicted = []
        with self._lock:
            self._entries[key] = ctx
            ctx._lease()
            while len(self._entries) > self.max_cached:
                evicted.append(self._entries.pop(next(iter(self._entries))))
            self._loading.pop(key).set()
        for old in evicted:
            old._retire()
        return ctx

    def clear(self):
        with self._lock:
            gone = list(self._entries.values())
            self._entries.clear()
        for ctx in gone:
            ctx._retire()

    def __len__(self):
        with self._lock:
            return len(self._entries)

    def __contains__(self, key):
        with self._lock:
            return key in self._entries


class LibsnarkReduction(R1CSToQAP):
    """forks/groth16/src/r1cs_to_qap.rs:100-226, the default QAP type (lib.rs:55), on the GPU.  The reference's call
    is stateless; the matrices' device copy is kept per matrices CONTENT (shape + digest of every term, at most MAX_CACHED
    resident, least recently used first out) so that repeated calls only move the assignment."""
    MAX_CACHED = 4
    _cache = _ResidentCache(MAX_CACHED)

    @classmethod
    def witness_map_from_matrices(cls, matrices: ConstraintMatrices, num_inputs: int, num_constraints: int,
                                  full_assignment) -> np.ndarray:
        """r1cs_to_qap.rs:150-213 -> the coefficients of h, domain_size x 32 B canonical.  Raises CrescentGpuError
        with code CG_ERR_POLY_DEGREE_TOO_LARGE where the reference returns PolynomialDegreeTooLarge (:156-157)."""
        if num_inputs != matrices.num_instance_variables or num_constraints != matrices.num_constraints:
            raise ValueError("num_inputs/num_constraints disagree with the matrices")
        cls._cache.max_cached = cls.MAX_CACHED
        ctx = cls._cache.lease(_matrices_key(matrices), lambda: QapContext(matrices))
        try:
            return ctx.witness_map(full_assignment)
        finally:
            ctx._unlease()

    @classmethod
    def clear_cache(cls):
        cls._cache.clear()


class Groth16:
    """forks/groth16/src/lib.rs:55-57 / prover.rs.  The reference's calls are stateless; here a loaded circuit (key
    tables + matrices in HBM) is kept per (key, matrices) CONTENT, since re-uploading a 0.6 GB key and redoing the 1.6 s
    change of basis for every proof would defeat the point - also when the caller parses fresh objects from the same
    files for every call, as `create_client_state` does (creds/src/lib.rs:258,268).  At most `MAX_CACHED` circuits stay
    resident; the least recently used one is retired first, and closed only once no thread is inside it."""
    MAX_CACHED = 4
    _cache = _ResidentCache(MAX_CACHED)

    @classmethod
    def _prover_for(cls, pk: ProvingKey, matrices: ConstraintMatrices) -> Prover:
        """a LEASED resident prover for this key and these matrices: the caller must `_unlease()` it"""
        cls._cache.max_cached = cls.MAX_CACHED
        # the load is STAGED (CG_FLAG_STAGED_LOAD): `Groth16::prove` is what create_client_state calls once per credential
        # (creds/src/lib.rs:281-283), so the first proof must not wait for tables that only pay off over many
        return cls._cache.lease((_pk_key(pk), _matrices_key(matrices)), lambda: Prover(pk, matrices, staged_load=True))

    @classmethod
    def create_proof_with_reduction_and_matrices(cls, pk: ProvingKey, r: int, s: int, matrices: ConstraintMatrices,
                                                 num_inputs: int, num_constraints: int, full_assignment) -> Proof:
        """prover.rs:26-51."""
        if num_inputs != matrices.num_instance_variables or num_constraints != matrices.num_constraints:
            raise ValueError("num_inputs/num_constraints disagree with the matrices")
        prover = cls._prover_for(pk, matrices)
        try:
            return prover.prove(full_assignment, r, s)
        finally:
            prover._unlease()

    @classmethod
    def create_proof_with_reduction(cls, circuit: "CircomCircuit", pk: ProvingKey, r: int, s: int) -> Proof:
        """prover.rs:177-221.  The reference synthesises the constraint system from the circuit on every call
        (circuit.rs:29-86) and extracts the matrices (r1cs_to_qap.rs:58-80); here the matrices ARE the circuit's
        R1CS (column = wire id, circuit.rs:61-67) and stay resident, so only the witness travels."""
        if circuit.witness is None:
            raise CrescentGpuError(-1, "AssignmentMissing: the circuit has no witness (SynthesisError::AssignmentMissing, circuit.rs:38-45)")
        cm = circuit.r1cs.matrices
        return cls.create_proof_with_reduction_and_matrices(pk, r, s, cm, cm.num_instance_variables, cm.num_constraints,
                                                            circuit.full_assignment())

    @classmethod
    def prove(cls, pk: ProvingKey, circuit: "CircomCircuit", rng) -> Proof:
        """`SNARK::prove` (lib.rs:76-82) -> create_random_proof_with_reduction (prover.rs:142-154): r and s are
        sampled from `rng` (any object with randrange, e.g. random.Random / random.SystemRandom), r first."""
        r = rng.randrange(FR_MODULUS)
        s = rng.randrange(FR_MODULUS)
        return cls.create_proof_with_reduction(circuit, pk, r, s)

    @classmethod
    def create_proof_no_zk(cls, circuit: "CircomCircuit", pk: ProvingKey) -> Proof:
        """prover.rs:160-173 (r = s = 0)."""
        return cls.create_proof_with_reduction(circuit, pk, 0, 0)

    @classmethod
    def clear_cache(cls):
        cls._cache.clear()


class CircomCircuit:
    """forks/circom-compat/src/circom/circuit.rs:11-27: an R1CS plus (optionally) the witness the WASM calculator
    produced for it.  `witness` is the full wire assignment in wire order (wire 0 = 1), canonical 32-byte scalars or
    Python ints; instance wires come first (circuit.rs:61-67), so it is also the prover's `full_assignment`."""

    def __init__(self, r1cs: "R1CSFile", witness=None):
        self.r1cs = r1cs
        self.witness = None
        if witness is not None:
            self.set_witness(witness)

    def set_witness(self, witness):
        if isinstance(witness, (list, tuple)):
            witness = scalars_to_array([int(x) % FR_MODULUS for x in witness])
        self.witness = _u8(witness, self.r1cs.num_variables * 32)

    def full_assignment(self) -> np.ndarray:
        return self.witness

    def get_public_inputs(self):
        """circuit.rs:18-26: wires 1 .. num_inputs-1 as integers (None without a witness)"""
        if self.witness is None:
            return None
        b = self.witness[32:32 * self.r1cs.num_inputs].tobytes()
        return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def generate_parameters_with_qap(matrices: ConstraintMatrices, alpha: int, beta: int, delta: int, tau: int) -> ProvingKey:
    """forks/groth16/src/generator.rs:50-228 with gamma = 1 and the standard generators (:28,:34-35), on the GPU."""
    l, m, M = matrices.num_instance_variables, matrices.num_constraints, matrices.num_variables
    D = 1
    while D < m + l:
        D <<= 1
    a = np.zeros(M * 64, np.uint8); b1 = np.zeros(M * 64, np.uint8); b2 = np.zeros(M * 128, np.uint8)
    h = np.zeros((D - 1) * 64, np.uint8); lq = np.zeros(max(M - l, 0) * 64, np.uint8) if M > l else np.zeros(0, np.uint8)
    gabc = np.zeros(l * 64, np.uint8); vkp = np.zeros(576, np.uint8)
    abc, _keep = matrices._c()
    tb, ab, bb, db = (_u8(fr_to_bytes(x)) for x in (tau, alpha, beta, delta))
    lq_buf = lq if lq.size else np.zeros(64, np.uint8)
    _check(lib().cg_setup(abc, l, m, M, _ptr(tb), _ptr(ab), _ptr(bb), _ptr(db), _ptr(a), _ptr(b1), _ptr(b2), _ptr(h),
                          _ptr(lq_buf), _ptr(gabc), _ptr(vkp)))
    vk = VerifyingKey(alpha_g1=vkp[0:64].copy(), beta_g2=vkp[192:320].copy(), gamma_g2=vkp[320:448].copy(),
                      delta_g1=vkp[128:192].copy(), delta_g2=vkp[448:576].copy(), gamma_abc_g1=gabc)
    return ProvingKey(vk=vk, beta_g1=vkp[64:128].copy(), delta_g1=vkp[128:192].copy(), a_query=a, b_g1_query=b1,
                      b_g2_query=b2, h_query=h, l_query=lq, coord_form=CG_FORM_CANONICAL)


# ------------------------------------------------------------------------------------------------
# unit-level operators
# ------------------------------------------------------------------------------------------------
def msm_bigint_g1(bases, scalars, coord_form: int = CG_FORM_CANONICAL, window_bits: int = 0) -> bytes:
    """<G1 as VariableBaseMSM>::msm_bigint (call sites prover.rs:66,74,266) -> affine canonical 64 B."""
    b, s = _u8(bases), _u8(scalars)
    out = np.zeros(64, np.uint8)
    _check(lib().cg_msm_g1(_ptr(b) if b.size else None, coord_form, b.size // 64, _ptr(s) if s.size else None,
                           s.size // 32, window_bits, _ptr(out)))
    return out.tobytes()


def msm_bigint_g2(bases, scalars, coord_form: int = CG_FORM_CANONICAL, window_bits: int = 0) -> bytes:
    b, s = _u8(bases), _u8(scalars)
    out = np.zeros(128, np.uint8)
    _check(lib().cg_msm_g2(_ptr(b) if b.size else None, coord_form, b.size // 128, _ptr(s) if s.size else None,
                           s.size // 32, window_bits, _ptr(out)))
    return out.tobytes()


def _ntt(data, inverse: bool, coset: bool) -> np.ndarray:
    a = _u8(data).copy()
    n = a.size // 32
    if n == 0 or n & (n - 1):
        raise ValueError("length must be a power of two")
    _check(lib().cg_ntt(_ptr(a), n.bit_length() - 1, 1 if inverse else 0, 1 if coset else 0))
    return a


def fft_in_place(data, coset: bool = False) -> np.ndarray:
    """EvaluationDomain::fft_in_place (coset=True: the `get_coset(F::GENERATOR)` domain, r1cs_to_qap.rs:182-185)."""
    return _ntt(data, False, coset)


def ifft_in_place(data, coset: bool = False) -> np.ndarray:
    """EvaluationDomain::ifft_in_place (r1cs_to_qap.rs:179-180,210)."""
    return _ntt(data, True, coset)


class MsmContext:
    """A fixed set of G1 (group=1) or G2 (group=2) bases resident on the GPU with its window tables; `run` is
    msm_bigint against them (cg_msm_load_* / cg_msm_run)."""

    def __init__(self, bases, group: int = 1, coord_form: int = CG_FORM_CANONICAL, window_bits: int = 0, device: int = -1):
        if group not in (1, 2):
            raise ValueError("group must be 1 or 2")
        self.group = group
        self.point_bytes = 64 * group
        b = _u8(bases)
        if b.size % self.point_bytes:
            raise ValueError("bases length is not a multiple of %d bytes" % self.point_bytes)
        self.n = b.size // self.point_bytes
        opt = _CgOptions(device=device, window_bits=window_bits)
        self._h = C.c_void_p()
        load = lib().cg_msm_load_g1 if group == 1 else lib().cg_msm_load_g2
        _check(load(C.byref(self._h), _ptr(b) if b.size else None, coord_form, self.n, C.byref(opt)))

    def run(self, scalars, timings: bool = False):
        s = _u8(scalars)
        return self._run(_ptr(s) if s.size else None, 0, s.size // 32, timings)

    def run_dev(self, d_ptr: int, n_scalars: int, timings: bool = False):
        """scalars already on this context's GPU (device pointer, n_scalars x 32 B canonical)"""
        return self._run(d_ptr, 1, n_scalars, timings)

    def _run(self, ptr, on_device, n, timings):
        out = np.zeros(self.point_bytes, np.uint8)
        tm = CgTimings()
        _check(lib().cg_msm_run(self._h, ptr, on_device, n, _ptr(out), C.byref(tm) if timings else None))
        return (out.tobytes(), tm.as_dict()) if timings else out.tobytes()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().cg_msm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NttContext:
    """A radix-2 domain of size 2^log_n resident on the GPU (cg_ntt_load / cg_ntt_run)."""

    def __init__(self, log_n: int, device: int = -1):
        self.log_n = log_n
        self._h = C.c_void_p()
        _check(lib().cg_ntt_load(C.byref(self._h), log_n, device))

    def run(self, data, inverse: bool = False, coset: bool = False) -> np.ndarray:
        a = _u8(data, 32 << self.log_n).copy()
        _check(lib().cg_ntt_run(self._h, _ptr(a), 0, 1 if inverse else 0, 1 if coset else 0, None))
        return a

    def run_dev(self, d_ptr: int, inverse: bool = False, coset: bool = False) -> float:
        """in place on device memory (2^log_n x 32 B canonical); returns the HIP-event time of the kernels in ms"""
        ms = C.c_float(0)
        _check(lib().cg_ntt_run(self._h, d_ptr, 1, 1 if inverse else 0, 1 if coset else 0, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().cg_ntt_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fixed_base_g1(scalars) -> bytes:
    """scalars[i]·G1 (FixedBase::msm, generator.rs:162-194) -> n x 64 B affine canonical"""
    s = _u8(scalars)
    out = np.zeros(s.size * 2, np.uint8)
    _check(lib().cg_fixed_base_g1(_ptr(s) if s.size else None, s.size // 32, _ptr(out) if out.size else None))
    return out.tobytes()


def fixed_base_g2(scalars) -> bytes:
    s = _u8(scalars)
    out = np.zeros(s.size * 4, np.uint8)
    _check(lib().cg_fixed_base_g2(_ptr(s) if s.size else None, s.size // 32, _ptr(out) if out.size else None))
    return out.tobytes()


def proving_key_from_bytes(data) -> Tuple["ProvingKey", int]:
    """`ProvingKey::deserialize_uncompressed_unchecked` (what creds/src/utils.rs:179-189 does to the head of
    prover_params.bin).  Returns (key with canonical packed arrays, bytes consumed)."""
    L = lib()
    buf = _u8(data)
    h = C.c_void_p()
    used = C.c_uint64()
    _check(L.cg_pk_parse(_ptr(buf), buf.size, C.byref(h), C.byref(used)))
    try:
        v = _CgProvingKey()
        g2 = C.c_void_p(); gabc = C.c_void_p(); ngabc = C.c_uint64()
        _check(L.cg_pk_get(h, C.byref(v), C.byref(g2), C.byref(gabc), C.byref(ngabc)))

        pk = _pk_from_view(v, g2, gabc, ngabc.value)
    finally:
        L.cg_pk_free(h)
    return pk, int(used.value)


def _arr(ptr, nbytes) -> np.ndarray:
    if not nbytes:
        return np.zeros(0, np.uint8)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)).copy()


def _pk_from_view(v: _CgProvingKey, gamma_g2, gamma_abc, n_gamma_abc) -> "ProvingKey":
    vk = VerifyingKey(alpha_g1=_arr(v.alpha_g1, 64), beta_g2=_arr(v.beta_g2, 128), gamma_g2=_arr(gamma_g2, 128),
                      delta_g1=_arr(v.delta_g1, 64), delta_g2=_arr(v.delta_g2, 128), gamma_abc_g1=_arr(gamma_abc, 64 * n_gamma_abc))
    return ProvingKey(vk=vk, beta_g1=_arr(v.beta_g1, 64), delta_g1=_arr(v.delta_g1, 64), a_query=_arr(v.a_query, 64 * v.a_len),
                      b_g1_query=_arr(v.b_g1_query, 64 * v.b_g1_len), b_g2_query=_arr(v.b_g2_query, 128 * v.b_g2_len),
                      h_query=_arr(v.h_query, 64 * v.h_len), l_query=_arr(v.l_query, 64 * v.l_len))


@dataclass
class ProverParams:
    """creds/src/lib.rs:58-63.  `groth16_pvk` and `vk_bytes` are the serialized PreparedVerifyingKey / VerifyingKey
    exactly as the file holds them: the prover passes them on to the ClientState (creds/src/lib.rs:292-299)."""
    groth16_params: ProvingKey
    groth16_pvk: bytes
    config_str: str
    vk_bytes: bytes = b""

    @staticmethod
    def from_bytes(data) -> "ProverParams":
        """`read_from_file::<ProverParams>` (creds/src/utils.rs:179-189, creds/src/lib.rs:268)"""
        L = lib()
        buf = _u8(data)
        h = C.c_void_p()
        _check(L.cg_prover_params_parse(_ptr(buf), buf.size, C.byref(h)))
        try:
            v = _CgProverParamsView()
            _check(L.cg_prover_params_get(h, C.byref(v)))
            pk = _pk_from_view(v.pk, v.gamma_g2, v.gamma_abc_g1, v.gamma_abc_len)
            return ProverParams(pk, _arr(v.pvk_bytes, v.pvk_len).tobytes(), _arr(v.config_str, v.config_len).tobytes().decode("utf-8"),
                                _arr(v.vk_bytes, v.vk_len).tobytes())
        finally:
            L.cg_prover_params_free(h)

    def to_bytes(self) -> bytes:
        """`write_to_file(&prover_params, ..)` (creds/src/utils.rs:140-152, creds/src/lib.rs:245-248)"""
        L = lib()
        pk = self.groth16_params
        cpk = pk._c()
        n = pk.vk.gamma_abc_g1.size // 64
        pvk = _u8(self.groth16_pvk)
        cfg = _u8(self.config_str.encode("utf-8"))
        size = int(L.cg_prover_params_serialized_size(C.byref(cpk), n, pvk.size, cfg.size))
        out = np.zeros(size, np.uint8)
        gabc = pk.vk.gamma_abc_g1 if n else np.zeros(64, np.uint8)
        _check(L.cg_prover_params_serialize(C.byref(cpk), _ptr(pk.vk.gamma_g2), _ptr(gabc), n, _ptr(pvk), pvk.size,
                                            _ptr(cfg) if cfg.size else None, cfg.size, _ptr(out), size))
        return out.tobytes()


@dataclass
class ClientState:
    """creds/src/groth16rand.rs:23-35.  vk / pvk travel as their serialized bytes (they come out of prover_params.bin);
    `committed_input_openings` as the serialized items (empty for a state fresh out of create_client_state)."""
    inputs: List[int]
    aux: Optional[str]
    proof: Proof
    vk: bytes
    pvk: bytes
    config_str: str
    credtype: str = "jwt"                       # groth16rand.rs:76
    input_com_randomness: Optional[int] = None
    committed_input_openings: bytes = b""
    n_openings: int = 0

    @staticmethod
    def new(inputs, aux, proof: Proof, vk: bytes, pvk: bytes, config_str: str) -> "ClientState":
        """ClientState::new (groth16rand.rs:60-80)"""
        return ClientState(list(inputs), aux, proof, vk, pvk, config_str)

    def to_bytes(self) -> bytes:
        """ClientState::write_to_file (groth16rand.rs:89-98)"""
        L = lib()
        v = _CgClientStateView()
        keep = []

        def put(field, length_field, data):
            a = _u8(data)
            keep.append(a)
            setattr(v, field, _ptr(a) if a.size else None)
            if length_field:
                setattr(v, length_field, a.size)
        put("inputs", None, scalars_to_array(self.inputs) if self.inputs else b"")
        v.n_inputs = len(self.inputs)
        v.has_aux = 0 if self.aux is None else 1
        put("aux", "aux_len", (self.aux or "").encode("utf-8"))
        put("proof", None, self.proof.data)
        put("vk_bytes", "vk_len", self.vk)
        put("pvk_bytes", "pvk_len", self.pvk)
        v.has_input_com_randomness = 0 if self.input_com_randomness is None else 1
        put("input_com_randomness", None, fr_to_bytes(self.input_com_randomness or 0))
        put("openings_bytes", "openings_len", self.committed_input_openings)
        v.n_openings = self.n_openings
        put("credtype", "credtype_len", self.credtype.encode("utf-8"))
        put("config_str", "config_len", self.config_str.encode("utf-8"))
        if len(self.proof.data) != 256:
            raise ValueError("proof must be 256 bytes")
        size = int(L.cg_client_state_serialized_size(C.byref(v)))
        out = np.zeros(size, np.uint8)
        _check(L.cg_client_state_serialize(C.byref(v), _ptr(out), size))
        return out.tobytes()

    @staticmethod
    def from_bytes(data) -> "ClientState":
        """ClientState::new_from_file (groth16rand.rs:82-87)"""
        L = lib()
        buf = _u8(data)
        h = C.c_void_p()
        _check(L.cg_client_state_parse(_ptr(buf), buf.size, C.byref(h)))
        try:
            v = _CgClientStateView()
            _check(L.cg_client_state_get(h, C.byref(v)))
            ib = _arr(v.inputs, 32 * v.n_inputs).tobytes()
            inputs = [int.from_bytes(ib[i:i + 32], "little") for i in range(0, len(ib), 32)]
            aux = _arr(v.aux, v.aux_len).tobytes().decode("utf-8") if v.has_aux else None
            rnd = int.from_bytes(_arr(v.input_com_randomness, 32).tobytes(), "little") if v.has_input_com_randomness else None
            return ClientState(inputs, aux, Proof(_arr(v.proof, 256).tobytes()), _arr(v.vk_bytes, v.vk_len).tobytes(),
                               _arr(v.pvk_bytes, v.pvk_len).tobytes(), _arr(v.config_str, v.config_len).tobytes().decode("utf-8"),
                               _arr(v.credtype, v.credtype_len).tobytes().decode("utf-8"), rnd,
                               _arr(v.openings_bytes, v.openings_len).tobytes(), int(v.n_openings))
        finally:
            L.cg_client_state_free(h)


class IOLocations:
    """creds/src/structs.rs:26-100: `io_locations.sym`, one `name,location` row per public input/output of the
    Groth16 circuit (location = wire index, so public input i of the proof is location - 1; creds/src/lib.rs:305-307)."""

    def __init__(self, io_data: str):
        self.public_io_locations = {}
        for line in io_data.splitlines():                       # new_from_str (structs.rs:48-68)
            parts = line.split(",")
            if len(parts) != 2:
                raise ValueError("Line %s in io_locations.sym is not formatted correctly! Found %d parts." % (line, len(parts)))
            if not parts[1].isdigit():
                raise ValueError("Line %s in io_locations.sym: location is not an unsigned integer" % line)
            self.public_io_locations[parts[0]] = int(parts[1])
        self.public_io_locations = dict(sorted(self.public_io_locations.items()))    # BTreeMap order

    @staticmethod
    def from_file(path: str) -> "IOLocations":
        with open(path) as f:
            return IOLocations(f.read())

    def get_io_location(self, key: str) -> int:
        if key not in self.public_io_locations:
            raise KeyError("Key %s not found in public_io_locations" % key)
        return self.public_io_locations[key]

    def get_public_key_indices(self) -> List[int]:
        """structs.rs:80-90: zero-based input positions of the issuer key limbs"""
        return sorted(v - 1 for k, v in self.public_io_locations.items() if k.startswith("modulus") or k.startswith("pubkey"))

    def get_all_names(self) -> List[str]:
        return list(self.public_io_locations.keys())


def create_client_state(r1cs_bytes, prover_params_bytes, witness, rng, prover_aux: Optional[str] = None,
                        credtype: str = "jwt", prover: Optional["Prover"] = None) -> "ClientState":
    """creds/src/lib.rs:255-301 without the witness generator: parse main_c.r1cs and prover_params.bin, prove with
    (r, s) drawn from `rng`, and assemble the ClientState the `show` step starts from.  `witness` is the full wire
    assignment the WASM calculator would have produced (wire 0 = 1).  The reference also verifies the proof against
    groth16_pvk.bin before returning (:286-290); that check is the caller's (the tests do it with the oracle)."""
    r1cs = R1CSFile(r1cs_bytes)
    pp = ProverParams.from_bytes(prover_params_bytes)
    circuit = CircomCircuit(r1cs, witness)
    if prover is not None:
        proof = prover.prove(circuit.full_assignment(), rng.randrange(FR_MODULUS), rng.randrange(FR_MODULUS))
    else:
        proof = Groth16.prove(pp.groth16_params, circuit, rng)
    cs = ClientState.new(circuit.get_public_inputs(), prover_aux, proof, pp.vk_bytes, pp.groth16_pvk, pp.config_str)
    cs.credtype = credtype
    return cs


def proving_key_to_bytes(pk: "ProvingKey") -> bytes:
    """`ProvingKey::serialize_uncompressed` (creds/src/utils.rs:140-152)."""
    L = lib()
    cpk = pk._c()
    n = pk.vk.gamma_abc_g1.size // 64
    size = int(L.cg_pk_serialized_size(C.byref(cpk), n))
    out = np.zeros(size, np.uint8)
    gabc = pk.vk.gamma_abc_g1 if n else np.zeros(64, np.uint8)
    _check(L.cg_pk_serialize(C.byref(cpk), _ptr(pk.vk.gamma_g2), _ptr(gabc), n, _ptr(out), size))
    return out.tobytes()


class R1CSFile:
    """forks/circom-compat/src/circom/r1cs_reader.rs:40-148 + R1CS::from (:26-38)."""

    def __init__(self, data: bytes):
        L = lib()
        buf = _u8(data)
        h = C.c_void_p()
        _check(L.cg_r1cs_parse(_ptr(buf), buf.size, C.byref(h)))
        try:
            hdr = _CgR1csHeader()
            abc = (_CgCsr * 3)()
            wm = C.c_void_p()
            _check(L.cg_r1cs_get(h, C.byref(hdr), abc, C.byref(wm)))
            self.header = {k: getattr(hdr, k) for k, _ in hdr._fields_}
            n_c = hdr.n_constraints
            mats = []
            for k in range(3):
                nnz = abc[k].nnz
                rp = np.ctypeslib.as_array(C.cast(abc[k].row_ptr, C.POINTER(C.c_uint64)), shape=(n_c + 1,)).copy()
                if nnz:
                    col = np.ctypeslib.as_array(C.cast(abc[k].col, C.POINTER(C.c_uint32)), shape=(nnz,)).copy()
                    coeff = np.ctypeslib.as_array(C.cast(abc[k].coeff, C.POINTER(C.c_uint8)), shape=(nnz * 32,)).copy()
                else:
                    col = np.zeros(0, np.uint32); coeff = np.zeros(0, np.uint8)
                mats.append(_Csr(rp, col, coeff))
            self.wire_mapping = np.ctypeslib.as_array(C.cast(wm, C.POINTER(C.c_uint64)), shape=(hdr.n_wires,)).copy()
            self.num_inputs = int(hdr.num_inputs)
            self.num_variables = int(hdr.num_variables)
            self.num_aux = self.num_variables - self.num_inputs
            self.matrices = ConstraintMatrices(mats[0], mats[1], mats[2], self.num_inputs, self.num_aux, n_c)
        finally:
            L.cg_r1cs_free(h)

    def constraint(self, i: int):
        """(A_i, B_i, C_i) as lists of (wire, coeff:int) — the reference's `Constraints<E>` tuple."""
        out = []
        for m in (self.matrices.a, self.matrices.b, self.matrices.c):
            lo, hi = int(m.row_ptr[i]), int(m.row_ptr[i + 1])
            out.append([(int(m.col[t]), int.from_bytes(m.coeff[32 * t:32 * t + 32].tobytes(), "little")) for t in range(lo, hi)])
        return tuple(out)
