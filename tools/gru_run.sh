set -x
mkdir -p gpurun_out/gru
python -m pytest tests/test_gpu_gru.py -x -q 2>&1 | tail -15 > gpurun_out/gru/test_gru.log
python -m pytest tests/test_gpu_parity.py -x -q -k "a2c_text or tower or txt" 2>&1 | tail -8 > gpurun_out/gru/test_golden.log
python tools/gru_probe.py > gpurun_out/gru/probe.json 2> gpurun_out/gru/probe.err
CFL_NO_GRU_FUSED=1 python bench.py --config 2 --round none --steps 30 --warmup 5 > gpurun_out/gru/c2_off.json 2> gpurun_out/gru/c2_off.err
python bench.py --config 2 --round none --steps 30 --warmup 5 > gpurun_out/gru/c2_on.json 2> gpurun_out/gru/c2_on.err
cat gpurun_out/gru/test_gru.log gpurun_out/gru/test_golden.log gpurun_out/gru/probe.json
