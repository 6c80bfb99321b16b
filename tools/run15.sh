mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q -k "remap or u8 or uint8 or whole_batch or masks or full_size" 2>&1 | tail -4
python tools/u8_bench.py > gpurun_out/r3/u8_bench.txt 2>&1
