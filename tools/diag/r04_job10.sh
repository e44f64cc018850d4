timeout -k 5 60 tools/micro/ordered_splat_check > gpurun_out/r04j_ordered_splat_check.log 2>&1; echo "rc=$?" >> gpurun_out/r04j_ordered_splat_check.log
cat gpurun_out/r04j_ordered_splat_check.log
