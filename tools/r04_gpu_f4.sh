#!/bin/bash
cd /root/repo
python -m pytest tests/test_hip_winoconv.py tests/test_hip_wino2conv.py tests/test_hip_gemm_x3.py -q -x 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | cut -c1-300
