C=staticfusion_amd/csrc
( timeout -k 5 900 python tools/diag/attribution_hunt.py --first 60000 --count 3000 --size 320x240 --libs reforder=$C/libsf_hip_reforder.so --json gpurun_out/r04u_hunt_ro_160x120_s60000_n3000.json 2>&1 | grep -v "^seed.*product" | tail -12 ) > gpurun_out/r04u_hunt_ro_160.log 2>&1; cat gpurun_out/r04u_hunt_ro_160.log | cut -c1-250
