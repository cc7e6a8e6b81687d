set -x
mkdir -p gpurun_out/gru
python -m pytest tests/test_gpu_gru.py tests/test_gpu_framework.py tests/test_gpu_parity.py -x -q -k "gru or bigru or hip_graph or recurrences or embedding or a2c_text or tower" 2>&1 | tail -15 > gpurun_out/gru/test_gru.log
python tools/gru_probe.py > gpurun_out/gru/probe.json 2> gpurun_out/gru/probe.err
python bench.py --config 2 --steps 30 --warmup 5 > gpurun_out/gru/c2_full.json 2> gpurun_out/gru/c2_full.err
cat gpurun_out/gru/test_gru.log gpurun_out/gru/probe.json
tail -3 gpurun_out/gru/c2_full.err
