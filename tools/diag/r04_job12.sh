timeout -k 5 60 tools/micro/ordered_splat_check > gpurun_out/r04l_ordered_splat_check.log 2>&1; echo "rc=$?" >> gpurun_out/r04l_ordered_splat_check.log
cat gpurun_out/r04l_ordered_splat_check.log
( timeout -k 5 120 python -m pytest tests/test_gpu_reference_order.py -m gpu -q -x -k "odd_geometry" 2>&1 | head -12 ) > gpurun_out/r04l_odd.log 2>&1
cat gpurun_out/r04l_odd.log | cut -c1-200
