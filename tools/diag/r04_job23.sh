C=staticfusion_amd/csrc
( timeout -k 5 200 python tools/diag/ro_diff.py 61826 2>&1 | tail -30 ) > gpurun_out/r04v_ro_diff_61826.log 2>&1; cut -c1-300 gpurun_out/r04v_ro_diff_61826.log
( timeout -k 5 600 python -m pytest tests/test_gpu_reference_order.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r04v_ro_tests.log 2>&1; cat gpurun_out/r04v_ro_tests.log
( timeout -k 5 900 python tools/diag/attribution_hunt.py --first 60000 --count 3000 --size 320x240 --libs reforder=$C/libsf_hip_reforder.so --json gpurun_out/r04v_hunt_ro_160x120_s60000_n3000.json 2>&1 | tail -4 ) > gpurun_out/r04v_hunt_ro_160.log 2>&1; cut -c1-200 gpurun_out/r04v_hunt_ro_160.log
