#!/bin/bash
# where does k_sort_lo_staged16's write amplification come from (1.55 GB written per dispatch for 0.58 GB of 4-byte entries)?
# One MSM-only loop (k = 22, 3 columns) under three counter passes — write requests from the L2 to the fabric split by size (32 B vs 64 B), WRITE_SIZE
# itself — and a kernel trace, for the default tile (8192 pairs: 4 entries = 16 bytes per (tile, bin) run) and for 16384-pair tiles (32-byte runs).
#   gpurun -- bash tools/sort_write_probe.sh r06_sortw
tag=${1:-r06_sortw}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
for v in tile8k tile16k; do
  if [ $v = tile16k ]; then export ZKHIP_SORT_TILE=16384 ZKHIP_SORT_WIDE=2; else unset ZKHIP_SORT_TILE ZKHIP_SORT_WIDE; fi
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/wrreq_$v -- python3 $root/tools/msm_only.py 22 3 > $out/wrreq_$v.txt 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/wsize_$v -- python3 $root/tools/msm_only.py 22 3 > $out/wsize_$v.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fsize_$v -- python3 $root/tools/msm_only.py 22 3 > $out/fsize_$v.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -- python3 $root/tools/msm_only.py 22 3 > $out/stats_$v.txt 2>&1
done
cd $root
python3 tools/sort_write_reduce.py $out | tee $out/summary.txt
