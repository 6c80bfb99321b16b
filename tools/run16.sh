python -m pytest tests -m gpu -x -q -k "resize or float64 or save_warped or warp_image or config1" 2>&1 | tail -15
