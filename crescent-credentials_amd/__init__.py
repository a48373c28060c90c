"""crescent-credentials_amd — MI355X-native Groth16 prove path for Crescent (BN254).

The product is the C-ABI shared library `libcrescent_gpu.so` (include/crescent_gpu.h), built from
hand-written HIP for gfx950 under `csrc/`.  This Python package is the thin host-side mirror of the
reference's `forks/groth16` interface used by the tests and the benchmark; it performs no
arithmetic itself and raises if the HIP library is missing (there is no CPU fallback).
"""
from .api import (  # noqa: F401
    CrescentGpuError,
    HostBuffer,
    host_register,
    host_unregister,
    probe_shader_clock,
    set_device,
    CircomCircuit,
    ClientState,
    IOLocations,
    ProverParams,
    create_client_state,
    ConstraintMatrices,
    Groth16,
    LibsnarkReduction,
    QapContext,
    R1CSToQAP,
    MsmContext,
    NttContext,
    Proof,
    Prover,
    ProvingKey,
    R1CSFile,
    proving_key_from_bytes,
    proving_key_to_bytes,
    VerifyingKey,
    fft_in_place,
    fixed_base_g1,
    fixed_base_g2,
    ifft_in_place,
    generate_parameters_with_qap,
    lib,
    library_path,
    msm_bigint_g1,
    msm_bigint_g2,
)
