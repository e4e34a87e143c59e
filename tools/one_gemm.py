#!/usr/bin/env python3
"""Run ONE smx_gemm shape a few times (profiling target).  usage: one_gemm.py LAYOUT N K M [epi]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
layout, N, K, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
run(N, K, M, layout, epi=sys.argv[5] if len(sys.argv) > 5 else "swishz")
