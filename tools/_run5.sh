for lib in halo2-zkcert_amd/libzkhip.so tools/ab/libzkhip_notop32.so; do
echo "=== $lib"
ZKHIP_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_polyops.py -q 2>&1 | tail -12
done
