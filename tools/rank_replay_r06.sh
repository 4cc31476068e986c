#!/bin/bash
# Round 6: the single-rank replay as a BOUND instead of a point.  Every rank of the sharded k = 22 proof (N = 8: all eight; N = 4: all
# four; N = 2: both) and every rank of the N = 8 chain is replayed alone on one GPU (bench.py --replay-rank R --of N; tools/replay_rccl), each with its
# per-exchange timeline recorded (zkhip_comm_trace), at four wire points: none, 10 us + 100 GB/s, 20 us + 50 GB/s, 40 us + 25 GB/s per link.
# Writes gpurun_out/<tag>/detail/<label>.json (bench.py's detail object per run); tools/install_rank_replay_r06.py reduces them to profiles/<tag>_rank_replay.json.
#   gpurun --timeout 2400 -- bash tools/rank_replay_r06.sh r06
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out/detail
: > $out/rank_replay.err
run() {   # label, args...
  label=$1; shift
  python bench.py --no-cpu-baseline "$@" --detail-out $out/detail/$label.json > $out/$label.line 2>> $out/rank_replay.err
  echo "$label: $(python -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"])' $out/$label.line 2>&1 | tail -1)"
  rm -f $out/$label.line
}
S="--steps 8 --warmup 2 --no-other-configs"
run single_k22 $S
wire() { case $1 in w0) echo "";; w10) echo "--replay-latency-us 10 --replay-link-gbs 100";; w20) echo "--replay-latency-us 20 --replay-link-gbs 50";; w40) echo "--replay-latency-us 40 --replay-link-gbs 25";; esac; }
for w in w0 w10 w20 w40; do
  for r in 0 1 2 3 4 5 6 7; do run k22_of8_rank${r}_$w $S --replay-rank $r --of 8 $(wire $w); done
done
for w in w0 w20; do
  for r in 0 1 2 3; do run k22_of4_rank${r}_$w $S --replay-rank $r --of 4 $(wire $w); done
  for r in 0 1; do run k22_of2_rank${r}_$w $S --replay-rank $r --of 2 $(wire $w); done
done
C="--chain --steps 4 --warmup 1"
run chain_single $C
for w in w0 w10 w20 w40; do
  for r in 0 1 2 3 4 5 6 7; do run chain_of8_rank${r}_$w $C --leaf-groups --replay-rank $r --of 8 $(wire $w); done
  # one rank per leaf (the default since round 6): rank 0 = an RSA leaf, rank 1 = a SHA leaf alone, rank 6 = no leaf; the aggregation phase is the grouped rows' own
  for r in 0 1 6; do run chain_nogroups_of8_rank${r}_$w $C --replay-rank $r --of 8 $(wire $w); done
done
