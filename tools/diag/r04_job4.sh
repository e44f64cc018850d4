set -x
C=staticfusion_amd/csrc
LIBS="reforder=$C/libsf_hip_reforder.so,product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so"
for n in splat_int splat_int_fine splat_int_coarse rows_fact rows_fact_fma fast_weights p1_fp32 labsum_int jacobi_rr init_res behind all_shortcuts; do LIBS="$LIBS,$n=$C/libsf_hip_ro_$n.so"; done
timeout 900 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 \
  --libs reforder=$C/libsf_hip_reforder.so,splat_int_fine=$C/libsf_hip_ro_splat_int_fine.so,splat_int_coarse=$C/libsf_hip_ro_splat_int_coarse.so,splat_int=$C/libsf_hip_ro_splat_int.so \
  --json gpurun_out/r04d_attr_splat_levels_qvga_s8000_n600.json > gpurun_out/r04d_attr_splat_levels_qvga_s8000_n600.log 2>&1
tail -8 gpurun_out/r04d_attr_splat_levels_qvga_s8000_n600.log
timeout 2400 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 --libs $LIBS \
  --json gpurun_out/r04d_attr_160x120_s50000_n5000.json > gpurun_out/r04d_attr_160x120_s50000_n5000.log 2>&1
tail -20 gpurun_out/r04d_attr_160x120_s50000_n5000.log
