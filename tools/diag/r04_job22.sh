( timeout -k 5 300 python tools/diag/ro_diff.py 61826 2>&1 | tail -60 ) > gpurun_out/r04u_ro_diff_61826.log 2>&1; cut -c1-400 gpurun_out/r04u_ro_diff_61826.log
