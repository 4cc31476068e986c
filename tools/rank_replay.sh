#!/bin/bash
# One rank's share of the sharded proofs, timed alone on one GPU (bench.py --replay-rank; tools/replay_rccl): the k = 22 headline at N = 2 / 4 / 8
# (the chain: tools/rank_replay_chain.sh), without and with a modelled wire, beside the single-GPU line of the same box.  Writes gpurun_out/<tag>/rank_replay.jsonl
# (one bench line per run, prefixed by its label); tools/install_rank_replay.py reduces it to profiles/<tag>_rank_replay.json.
#   gpurun --timeout 2400 -- bash tools/rank_replay.sh r05
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
: > $out/rank_replay.jsonl
run() {   # label, args...
  label=$1; shift
  line=$(python bench.py --no-cpu-baseline "$@" 2>> $out/rank_replay.err | tail -1)
  echo "{\"label\": \"$label\", \"line\": $line}" >> $out/rank_replay.jsonl
  echo "$label: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], (d.get("replay") or {}).get("exchanges_per_step"))')"
}
S="--steps 10 --warmup 2 --no-other-configs"
run single_k22 $S
for n in 2 4 8; do
  run k22_rank0_of$n $S --replay-rank 0 --of $n
  run k22_rank0_of${n}_wire20us_50GBs $S --replay-rank 0 --of $n --replay-latency-us 20 --replay-link-gbs 50
done
ZKHIP_COMM_BULK=0 run k22_rank0_of8_wire20us_50GBs_one_communicator $S --replay-rank 0 --of 8 --replay-latency-us 20 --replay-link-gbs 50
ZKHIP_COMM_BULK=0 run k22_rank0_of4_wire20us_50GBs_one_communicator $S --replay-rank 0 --of 4 --replay-latency-us 20 --replay-link-gbs 50
run k22_rank7_of8 $S --replay-rank 7 --of 8
run k22_rank3_of8 $S --replay-rank 3 --of 8
run k22_rank0_of2_columns $S --replay-rank 0 --of 2 --shard columns
run k22_rank0_of2_columns_wire20us_50GBs $S --replay-rank 0 --of 2 --shard columns --replay-latency-us 20 --replay-link-gbs 50
run k22_rank0_of8_allgather $S --replay-rank 0 --of 8 --row-sharded 0
run k22_rank0_of8_columns $S --replay-rank 0 --of 8 --shard columns
# the shard tables' window width at N = 8 (2^19 points per rank: c = 17 by default)
for c in 15 16; do run k22_rank0_of8_c$c $S --replay-rank 0 --of 8 --msm-c $c; done
run sha19_rank0_of2 --config sha19 $S --replay-rank 0 --of 2
run sha19_rank0_of3 --config sha19 $S --replay-rank 0 --of 3
# k = 19 / 17 sharded by column (whole tables on every rank)
run sha19_single --config sha19 $S
run sha19_rank0_of8 --config sha19 $S --replay-rank 0 --of 8
run rsa17_single --config rsa17 --steps 30 --warmup 3 --no-other-configs
run rsa17_rank0_of8 --config rsa17 --steps 30 --warmup 3 --no-other-configs --replay-rank 0 --of 8
