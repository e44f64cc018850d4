set -x
C=staticfusion_amd/csrc
# K-means v2: labels bit-exact on all three builds (the cluster build has its own K-means: unchanged), then the whole parity file
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_reference_order.py -m gpu -q 2>&1 | tail -15
SF_TEST_VARIANTS=throughput timeout 1500 python -m pytest tests/test_gpu_parity_hunt.py -m gpu -q -s 2>&1 | grep -v "^seed\|^QVGA seed" | tail -25
# A/B: v1 (round 3) against v2 Lloyd pass
timeout 1200 bash tools/ab_compare.sh libsf_hip_kmv1.so libsf_hip.so 3 5120 kmeans km_assign 2>&1 | tail -16
# what an ordered coarse splat would buy the product
timeout 1500 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 \
  --libs all_shortcuts=$C/libsf_hip_ro_all_shortcuts.so,all_but_coarse_splat=$C/libsf_hip_ro_all_but_coarse_splat.so \
  --json gpurun_out/r04e_attr_coarse_splat_160x120_s50000_n5000.json > gpurun_out/r04e_attr_coarse_splat_160x120_s50000_n5000.log 2>&1
tail -5 gpurun_out/r04e_attr_coarse_splat_160x120_s50000_n5000.log
