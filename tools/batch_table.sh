#!/usr/bin/env bash
# ms per step and encoder frames/s over the batch size (C2b training step and forward pass, T = 500; C2a at the recipe batch):
#   bash tools/batch_table.sh        (hipGraph replay below 40 000 frames per step, as bench.py chooses)
one() { python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %8.3f ms  %10.0f frames/s   %s' % ('$*', d['ms_per_step'], d['value'], d['config'].get('launch','')[:40]))"; }
echo "# C2b (12 L, d_model 256), bf16 operands, fp32 residual stream, T = 500"
for b in 1 2 4 8 16 32 64 128; do one --batch $b; done
for b in 1 4 8 32 128; do one --mode forward --batch $b; done
echo "# C2a (12 L, d_model 512), the recipe's 150 s batch and single utterances"
one --config c2a --batch 10 --frames 375
one --config c2a --batch 1 --frames 375
one --config c2a --mode forward --batch 1 --frames 375
echo "# the recipe's optimizer step: grad_accumulation_factor micro-batches of 10 x 375, accumulated one after the other / fused into one batch"
for g in 2 4 8; do one --config c2a --batch 10 --frames 375 --grad-accum $g --accum sequential; one --config c2a --batch 10 --frames 375 --grad-accum $g --accum fused; done
