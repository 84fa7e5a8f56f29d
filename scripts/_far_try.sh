set -e
mkdir -p gpurun_out/far
python -m pytest tests/test_gpu_moment_kernel.py tests/test_gpu_batch.py tests/test_gpu_pipeline.py -q -x -m gpu > gpurun_out/far/tests.log 2>&1
for t in 512 1024 256; do
  GRT_FAR_TILE=$t python bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 2 > gpurun_out/far/bench_$t.json 2> gpurun_out/far/bench_$t.err
done
