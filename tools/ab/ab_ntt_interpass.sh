# what the inter-pass Cooley-Tukey product of the non-final NTT passes costs — timing-only builds of ntt.hip
# (tools/ab/ntt_interpass.patch: -DNTT_IP_NOTAB no table read (the product by a computed constant stays), -DNTT_IP_NOMUL the table read
# stays but the product is an xor, -DNTT_IP_NONE neither; results WRONG by construction) against the shipped library, isolated and in the
# k = 22 proof, shipped first and last.  Build on the CPU box first (see profiles/r05_ntt_coset_passes.md), then
#   gpurun -- bash tools/ab/ab_ntt_interpass.sh
R=${GRAFT_REPO_ROOT:-.}
for v in shipped NOTAB NOMUL NONE shipped; do
  if [ $v = shipped ]; then export ZKHIP_LIB=$R/halo2-zkcert_amd/libzkhip.so; else export ZKHIP_LIB=$R/tools/ab/ntt_ip_$v.so; fi
  echo "== $v"
  python3 $R/tools/kernel_bench.py --skip-msm --ks 22 2>&1 | grep -E "NTT|extended|coset"
  python3 $R/bench.py --no-other-configs --no-cpu-baseline --no-h2d --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['configs']['agg22']; print('proof ms', j['ms_per_step'], {k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if k.startswith('ntt')}, 'isolated 8 cols onto cosets ms', c['rooflines']['ntt']['avg_launch_ms']*c['rooflines']['ntt']['algorithmic_bytes_per_launch'] and round(64.0*8*3*(1<<22)/c['rooflines']['ntt']['achieved']/1e6,3))"
done
