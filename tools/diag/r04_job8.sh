C=staticfusion_amd/csrc
( timeout -k 5 150 python -m pytest tests/test_gpu_reference_order.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r04h_reforder_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04h_reforder_tests.log
cat gpurun_out/r04h_reforder_tests.log
( SF_TEST_VARIANTS=throughput timeout -k 5 200 python -m pytest tests/test_gpu_parity.py tests/test_multi_frame.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r04h_product_tests.log 2>&1
cat gpurun_out/r04h_product_tests.log
