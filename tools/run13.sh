mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q -k "whole_batch" --durations=5 2>&1 | tail -25
