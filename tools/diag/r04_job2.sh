set -x
export SF_TEST_VARIANTS=throughput
timeout 600 python tools/diag/attribution_hunt.py --first 8000 --count 24 --size 640x480 \
  --libs reforder=staticfusion_amd/csrc/libsf_hip_reforder.so,product=staticfusion_amd/csrc/libsf_hip.so \
  --json gpurun_out/r04b_attr_qvga_s8000_n24.json 2>&1 | tail -40
SF_HIP_LIB=$PWD/staticfusion_amd/csrc/libsf_hip_reforder.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_rules.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -30
