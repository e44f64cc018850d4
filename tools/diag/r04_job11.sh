C=staticfusion_amd/csrc
run() { name=$1; shift; ( "$@" ) > gpurun_out/r04k_$name.log 2>&1; echo "rc=$?" >> gpurun_out/r04k_$name.log; tail -${TAIL:-8} gpurun_out/r04k_$name.log | cut -c1-220; }
run reforder_tests timeout -k 5 200 python -m pytest tests/test_gpu_reference_order.py -m gpu -q
SF_TEST_VARIANTS=throughput run hunt_subset timeout -k 5 400 python -m pytest tests/test_gpu_parity_hunt.py -m gpu -q -k "reference_order or excursion" -s
TAIL=16 run ab timeout -k 5 600 bash tools/ab_compare.sh libsf_hip_nocoarse.so libsf_hip.so 3 5120 warp linearise
TAIL=25 run full_suite timeout -k 5 1500 python -m pytest tests -m gpu -q
