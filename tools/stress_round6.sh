#!/bin/bash
# randomised parity stress after round 6's changes: msm_local split into partials + tail (every MSM of every proof goes through it), the pipelined host-slice MSM,
# zkhip_comm_init's refactored self-checks (readiness agreed before the split is used; comm_bulk now opt-in), the exchange trace
#   gpurun -- bash tools/stress_round6.sh
P="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
F="ZKHIP_RCCL_LIB=tests/fake_rccl/libfake_rccl.so ZKHIP_COMM_TRANSPORT=rccl ZKFAKE_RCCL_SLOT_MB=8"
env $F ZKHIP_COMM_BULK=1 $P --nproc-per-node=4 --master-port 29577 tools/stress_dist.py --seconds 120 2>&1 | grep -v "^\[W\|^W1" | tail -2
env $F ZKHIP_COMM_BULK=1 $P --nproc-per-node=8 --master-port 29578 tools/stress_dist.py --seconds 120 --kmax 11 2>&1 | grep -v "^\[W\|^W1" | tail -2
env $F $P --nproc-per-node=4 --master-port 29579 tools/stress_dist.py --seconds 60 2>&1 | grep -v "^\[W\|^W1" | tail -2
$P --nproc-per-node=3 --master-port 29580 tools/stress_dist.py --seconds 60 2>&1 | grep -v "^\[W\|^W1" | tail -2
python tools/stress_proofs.py --seconds 150 --kmin 6 --kmax 13 2>&1 | tail -1
python tools/stress.py --seconds 120 --kmax 15 2>&1 | tail -1
python tools/host_msm_stress.py --seconds 60 --kmin 16 --kmax 21 2>&1 | tail -1
