set -x
C=staticfusion_amd/csrc
timeout 600 python -m pytest tests/test_gpu_reference_order.py -m gpu -q 2>&1 | tail -8
SF_TEST_VARIANTS=throughput timeout 900 python -m pytest tests/test_gpu_parity_hunt.py -m gpu -q -k "reference_order or excursion" -s 2>&1 | grep -v "^seed\|^QVGA seed" | tail -12
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25
timeout 900 bash tools/ab_compare.sh libsf_hip_nocoarse.so libsf_hip.so 3 5120 warp linearise 2>&1 | tail -4
timeout 900 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 \
  --libs product=$C/libsf_hip.so,reforder=$C/libsf_hip_reforder.so,splat_int_fine=$C/libsf_hip_ro_splat_int_fine.so \
  --json gpurun_out/r04g_attr_product_160x120_s50000_n5000.json > gpurun_out/r04g_attr_product_160x120_s50000_n5000.log 2>&1
tail -5 gpurun_out/r04g_attr_product_160x120_s50000_n5000.log
