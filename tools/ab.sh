#!/bin/bash
# usage: tools/ab.sh name1 name2 ...   (variants in scratch/variants/lib_<name>.so; "base" = in-tree lib)
# AB_ARGS="--samples 4096" passes extra bench.py options.
# Interleaved A/B: two rounds over all variants in one box session (box-to-box variance is ~4 %).
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset OCTPIPE_LIB; else export OCTPIPE_LIB=$PWD/scratch/variants/lib_$v.so; fi
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), 'M A-scans/s  kernel_ms', round(d['roofline']['kernel_ms'],4))"
done; done
