( timeout -k 5 90 python tools/diag/odd_case.py libsf_hip_reforder.so 3 1 frame 2>&1 | grep -v "^  File\|^$\|Extension\|Thread" | head -12 ) > gpurun_out/r04p_odd_case.log 2>&1
cat gpurun_out/r04p_odd_case.log | cut -c1-200
( timeout -k 5 300 python -m pytest tests/test_gpu_reference_order.py -m gpu -q 2>&1 | grep -v "^  File\|^$\|Extension\|Thread" | tail -8 ) > gpurun_out/r04p_all.log 2>&1
cat gpurun_out/r04p_all.log | cut -c1-200
