#!/bin/bash
# the chain's rows of tools/rank_replay.sh alone (BASELINE configs[4]; N = 8: rank 0 = RSA leaf, rank 1 = head of a SHA leaf's pair, rank 4 = its second
# member, rank 6 = no leaf), with and without the leaf groups
tag=${1:-r05c}
out=gpurun_out/$tag
mkdir -p $out
: > $out/rank_replay.jsonl
run() {
  label=$1; shift
  line=$(python bench.py --no-cpu-baseline "$@" 2>> $out/rank_replay.err | tail -1)
  echo "{\"label\": \"$label\", \"line\": $line}" >> $out/rank_replay.jsonl
  echo "$label: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("phase_ms_per_step"))')"
}
C="--chain --steps 5 --warmup 1"
run chain_single $C
for r in 0 1 4 6; do run chain_rank${r}_of8 $C --replay-rank $r --of 8; done
run chain_rank1_of8_no_groups $C --replay-rank 1 --of 8 --no-leaf-groups
run chain_rank0_of4 $C --replay-rank 0 --of 4
run chain_rank1_of4 $C --replay-rank 1 --of 4
run chain_rank1_of6 $C --replay-rank 1 --of 6
