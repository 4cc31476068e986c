#!/bin/bash
# the bulk communicator (comm.hip, round 5) against everything on one communicator, by single-rank replay with the modelled wire (20 us + bytes / 50 GB/s per link):
# rank 0 of 8 and of 4, k = 22, alternating
#   gpurun -- bash tools/bulk_comm_ab.sh
for n in 8 4; do
  for b in 1 0 1 0; do
    line=$(ZKHIP_COMM_BULK=$b python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 --replay-rank 0 --of $n --replay-latency-us 20 --replay-link-gbs 50 2>/dev/null | tail -1)
    echo "N=$n comm_bulk=$b: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["comm"]["bulk_communicator"], d["replay"]["exchanges_per_step"])')"
  done
done
for b in 1 0; do
  line=$(ZKHIP_COMM_BULK=$b python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 --replay-rank 0 --of 8 2>/dev/null | tail -1)
  echo "N=8 no wire comm_bulk=$b: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
