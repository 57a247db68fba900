#!/bin/bash
cd /root/repo
python -m pytest tests/test_hip_bf16.py -q -m gpu -s -k "training_tracks" 2>&1 | grep -E "loss|passed|failed"
python -m pytest tests/test_hip_bf16.py tests/test_hip_model.py -q -m gpu 2>&1 | tail -3
bash tools/prof_step.sh r04d_rtod_bf16 --mode RtoD --dtype bf16 2>&1 | tail -12
