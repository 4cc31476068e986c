"""halo2-zkcert_amd: MI355X-native backend for the arithmetic under halo2_proofs::plonk::create_proof
(BN254 G1 MSM, Fr NTT, quotient sweep), behind the C ABI in include/zkhip.h.

The directory name is fixed by the build contract and is not a Python identifier; import it as
`halo2_zkcert_amd` through the shim module at the repo root.
"""
