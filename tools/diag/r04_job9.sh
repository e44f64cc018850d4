timeout -k 5 120 python tools/diag/warp_plane_check.py staticfusion_amd/csrc/libsf_hip_reforder.so staticfusion_amd/csrc/libsf_hip.so > gpurun_out/r04i_warp_plane_check.log 2>&1
cat gpurun_out/r04i_warp_plane_check.log
