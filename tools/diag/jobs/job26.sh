C=staticfusion_amd/csrc
( timeout -k 5 1500 python tools/diag/attribution_hunt.py --first 10000 --count 2400 --size 640x480 --libs product=$C/libsf_hip.so --json gpurun_out/r04x_hunt_fresh_qvga_s10000_n2400.json 2>&1 | tail -8 ) > gpurun_out/r04x_hunt_fresh_qvga_2400.log 2>&1; cut -c1-220 gpurun_out/r04x_hunt_fresh_qvga_2400.log
