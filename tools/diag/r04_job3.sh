set -x
C=staticfusion_amd/csrc
LIBS="reforder=$C/libsf_hip_reforder.so,product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so"
for n in splat_int splat_int_fastdiv rows_fact rows_fact_fma fast_weights p1_fp32 labsum_int jacobi_rr init_res behind all_shortcuts; do LIBS="$LIBS,$n=$C/libsf_hip_ro_$n.so"; done
timeout 2400 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 --libs $LIBS \
  --json gpurun_out/r04c_attr_qvga_s8000_n600.json > gpurun_out/r04c_attr_qvga_s8000_n600.log 2>&1
tail -20 gpurun_out/r04c_attr_qvga_s8000_n600.log
SF_TEST_VARIANTS=throughput,latency SF_HIP_LIB=$PWD/$C/libsf_hip_reforder.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_rules.py tests/test_golden.py tests/test_gpu_parity_hunt.py tests/test_multi_frame.py -m gpu -q 2>&1 | tail -30
