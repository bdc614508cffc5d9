#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/bench_configs.py --only cfg1,cfg1g 2>/dev/null
