mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tests/fuzz/fuzz_stages.py 20 31 2>&1 | tail -12
