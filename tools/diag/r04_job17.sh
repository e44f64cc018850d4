C=staticfusion_amd/csrc
run() { name=$1; shift; ( "$@" ) > gpurun_out/r04q_$name.log 2>&1; echo "rc=$?" >> gpurun_out/r04q_$name.log; tail -${TAIL:-8} gpurun_out/r04q_$name.log | cut -c1-220; }
TAIL=25 run full_suite timeout -k 5 1500 python -m pytest tests -m gpu -q
TAIL=6 run ab timeout -k 5 600 bash tools/ab_compare.sh libsf_hip_nocoarse.so libsf_hip.so 3 5120 warp linearise
TAIL=6 run hunt_160 timeout -k 5 600 python tools/diag/attribution_hunt.py --first 50000 --count 5000 --size 320x240 --libs product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so,before=$C/libsf_hip_nocoarse.so --json gpurun_out/r04q_attr_product_160x120_s50000_n5000.json
TAIL=6 run hunt_qvga timeout -k 5 600 python tools/diag/attribution_hunt.py --first 8000 --count 600 --size 640x480 --libs product=$C/libsf_hip.so,precise=$C/libsf_hip_precise.so,before=$C/libsf_hip_nocoarse.so,reforder=$C/libsf_hip_reforder.so --json gpurun_out/r04q_attr_product_qvga_s8000_n600.json
