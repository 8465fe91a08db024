for i in 1 2; do
echo "packed GELU:"; VARIANTS=4:0,0:0 python tools/mb_variants.py 2>&1 | grep "fc1"
echo "scalar GELU:"; CWM_HIP_LIB=$PWD/counterfactualworldmodels_amd/lib/libcwm_hip_scalar_gelu.so VARIANTS=4:0,0:0 python tools/mb_variants.py 2>&1 | grep "fc1"
done
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gelu or golden_block or linear" 2>&1 | tail -1
