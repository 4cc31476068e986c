#!/bin/bash
# randomised parity stress of the multi-rank path after round 5's changes (owners dealt across batches, the bulk communicator): sharded == single-GPU bytes
#   gpurun -- bash tools/stress_round5.sh
P="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
ZKHIP_RCCL_LIB=tests/fake_rccl/libfake_rccl.so ZKHIP_COMM_TRANSPORT=rccl ZKFAKE_RCCL_SLOT_MB=8 $P --nproc-per-node=4 --master-port 29577 tools/stress_dist.py --seconds 150 2>&1 | grep -v "^\[W\|^W1" | tail -2
ZKHIP_RCCL_LIB=tests/fake_rccl/libfake_rccl.so ZKHIP_COMM_TRANSPORT=rccl ZKFAKE_RCCL_SLOT_MB=8 $P --nproc-per-node=8 --master-port 29578 tools/stress_dist.py --seconds 150 --kmax 11 2>&1 | grep -v "^\[W\|^W1" | tail -2
$P --nproc-per-node=3 --master-port 29579 tools/stress_dist.py --seconds 100 2>&1 | grep -v "^\[W\|^W1" | tail -2
python tools/stress_proofs.py --seconds 120 --kmin 6 --kmax 12 2>&1 | tail -1
