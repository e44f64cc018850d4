set -x
C=staticfusion_amd/csrc
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
timeout 900 bash tools/ab_compare.sh libsf_hip_nocoarse.so libsf_hip.so 3 5120 warp linearise 2>&1 | tail -16
timeout 900 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 \
  --libs product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so,before=$C/libsf_hip_nocoarse.so \
  --json gpurun_out/r04f_attr_product_160x120_s50000_n5000.json > gpurun_out/r04f_attr_product_160x120_s50000_n5000.log 2>&1
tail -6 gpurun_out/r04f_attr_product_160x120_s50000_n5000.log
timeout 900 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 \
  --libs product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so,before=$C/libsf_hip_nocoarse.so \
  --json gpurun_out/r04f_attr_product_qvga_s8000_n600.json > gpurun_out/r04f_attr_product_qvga_s8000_n600.log 2>&1
tail -6 gpurun_out/r04f_attr_product_qvga_s8000_n600.log
