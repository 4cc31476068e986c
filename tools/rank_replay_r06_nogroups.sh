tag=r06; out=gpurun_out/$tag; mkdir -p $out/detail
run() { label=$1; shift; python bench.py --no-cpu-baseline "$@" --detail-out $out/detail/$label.json > $out/$label.line 2>> $out/rank_replay.err; echo "$label: $(tail -1 $out/$label.line | head -c 200)"; rm -f $out/$label.line; }
wire() { case $1 in w0) echo "";; w10) echo "--replay-latency-us 10 --replay-link-gbs 100";; w20) echo "--replay-latency-us 20 --replay-link-gbs 50";; w40) echo "--replay-latency-us 40 --replay-link-gbs 25";; esac; }
C="--chain --steps 4 --warmup 1"
for w in w0 w10 w20 w40; do for r in 0 1 6; do run chain_nogroups_of8_rank${r}_$w $C --replay-rank $r --of 8 $(wire $w); done; done
