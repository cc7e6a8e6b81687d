set -x
mkdir -p gpurun_out/gru
for h in 64 128; do python tools/gru_probe.py --hidden $h >> gpurun_out/gru/probe.jsonl 2>> gpurun_out/gru/probe.err; CFL_GRU_STREAM=1 python tools/gru_probe.py --hidden $h >> gpurun_out/gru/probe.jsonl 2>> gpurun_out/gru/probe.err; done
CFL_GRU_STREAM=1 python -m pytest tests/test_gpu_gru.py -x -q 2>&1 | tail -3
CFL_GRU_STREAM=1 python bench.py --config 2 --round none --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/gru/c2_stream.json 2> gpurun_out/gru/c2_stream.err
cat gpurun_out/gru/probe.jsonl
