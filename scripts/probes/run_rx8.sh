#!/bin/bash
O=gpurun_out; mkdir -p $O
for args in "1000 0" "8192 0" "8193 3" "100000 50" "100000 100" "3000000 1" "50000000 1" "4100000000 1"; do
  echo "== $args"; timeout -k 10 200 scripts/probes/rx8_bench.bin $args 2>&1 | tail -n 3 || exit 1
done > $O/r05_rx8_slot.txt 2>&1
cat $O/r05_rx8_slot.txt
