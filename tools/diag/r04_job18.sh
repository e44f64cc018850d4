C=staticfusion_amd/csrc
run() { name=$1; shift; ( "$@" ) > gpurun_out/r04r_$name.log 2>&1; echo "rc=$?" >> gpurun_out/r04r_$name.log; tail -${TAIL:-8} gpurun_out/r04r_$name.log | cut -c1-220; }
TAIL=5 run hunt_160 timeout -k 5 600 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 --libs coarse2048=$C/libsf_hip_coarse2048.so --json gpurun_out/r04r_attr_coarse2048_160x120_s50000_n5000.json
TAIL=5 run hunt_qvga timeout -k 5 600 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 --libs coarse2048=$C/libsf_hip_coarse2048.so,product=$C/libsf_hip.so --json gpurun_out/r04r_attr_coarse2048_qvga_s8000_n600.json
TAIL=4 run ab timeout -k 5 900 bash tools/ab_compare.sh libsf_hip_nocoarse.so libsf_hip_coarse2048.so 3 5120 warp
