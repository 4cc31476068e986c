# A/B of two builds of the library on the NTT only: isolated timings, rocprofv3 kernel-trace averages, and VALU / VMEM / wait counters
#   gpurun -- bash tools/ab/ab_ntt.sh     (base = tools/ab/base.so, new = halo2-zkcert_amd/libzkhip.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base new; do
  if [ $v = base ]; then export ZKHIP_LIB=$R/tools/ab/base.so; else export ZKHIP_LIB=$R/halo2-zkcert_amd/libzkhip.so; fi
  echo "== $v"
  python3 $R/tools/kernel_bench.py --skip-msm --ks 19,22 2>&1 | grep -E "NTT|extended"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abntt_$v -- python3 $R/tools/kernel_bench.py --skip-msm --ks 22 > /dev/null 2>&1
  f=$(ls $R/gpurun_out/abntt_$v/*/*kernel_stats.csv | head -1); grep -E "ntt|Name" $f | cut -c1-200
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/abnttpmc_$v -- python3 $R/tools/kernel_bench.py --skip-msm --ks 22 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/abnttpmc_$v/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "ntt" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[k]+=1
for k,c in acc.items(): print(k, n[k], {a: round(b/n[k]) for a,b in c.items()})
PY
done
