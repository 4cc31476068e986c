#!/bin/bash
# the chain rows of tools/rank_replay_r06.sh alone (re-run after a fix of the chain's phase clock: the traced steps were being added to the timed phases)
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out/detail
run() {
  label=$1; shift
  python bench.py --no-cpu-baseline "$@" --detail-out $out/detail/$label.json > $out/$label.line 2>> $out/rank_replay.err
  echo "$label: $(python -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"])' $out/$label.line 2>&1 | tail -1)"
  rm -f $out/$label.line
}
wire() { case $1 in w0) echo "";; w10) echo "--replay-latency-us 10 --replay-link-gbs 100";; w20) echo "--replay-latency-us 20 --replay-link-gbs 50";; w40) echo "--replay-latency-us 40 --replay-link-gbs 25";; esac; }
C="--chain --steps 4 --warmup 1"
run chain_single $C
for w in w0 w10 w20 w40; do
  for r in 0 1 2 3 4 5 6 7; do run chain_of8_rank${r}_$w $C --leaf-groups --replay-rank $r --of 8 $(wire $w); done
  # one rank per leaf (the default since round 6): rank 0 = an RSA leaf, rank 1 = a SHA leaf alone, rank 6 = no leaf; the aggregation phase is the grouped rows' own
  for r in 0 1 6; do run chain_nogroups_of8_rank${r}_$w $C --replay-rank $r --of 8 $(wire $w); done
done
