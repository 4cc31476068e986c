#!/bin/bash
# Host-side code of libzkhip.so under AddressSanitizer + UBSan, on the CPU (GPU ASan is not available on the pool):
# every translation unit is compiled with the sanitizers on the HOST side only (-Xarch_host), linked into build/asan/libzkhip_asan.so,
# and tools/asan_host_run.py drives what runs without a GPU: the three transcripts (BLAKE2b, Keccak, Poseidon incl. the Grain
# parameter generation), the host point helpers (batch normalisation, compression, Jacobian addition), option / communicator plumbing
# and the no-device error paths.      bash tools/asan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/asan
mkdir -p "$out"
rt=$(find /opt/rocm*/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
for f in ctx comm msm ntt sweep cosets srs_gen polyops permute shplonk prover transcript selftest; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
        -Wno-unused-function -c "$root/halo2-zkcert_amd/csrc/$f.hip" -o "$out/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -o "$out/libzkhip_asan.so" "$out"/*.o -ldl
ZKHIP_ASAN_LIB=$out/libzkhip_asan.so ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$rt python3 "$root/tools/asan_host_run.py"
