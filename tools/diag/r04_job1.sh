set -x
python -m pytest tests/test_gpu_parity_sweep.py -q -x -k "bench" 2>&1 | tail -5
for v in cluster latency throughput; do
  python bench.py --workload static --variant $v --batch 64 --steps 20 --warmup 5 --no-full-solver --no-sequences --no-cpu-baseline > gpurun_out/r04a_static_b64_$v.json 2> gpurun_out/r04a_static_b64_$v.err
  tail -c 600 gpurun_out/r04a_static_b64_$v.json
done
python bench.py --gpus 8 ; echo "rc=$?"
