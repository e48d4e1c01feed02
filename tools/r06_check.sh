#!/bin/bash
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_router_gpu.py tests/test_kv_move_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_mla_gpu.py -x -q -m gpu -k "timeout or metadata or parity" 2>&1 | tail -4
timeout 300 python tools/time_k1.py 2>&1 | tail -3
