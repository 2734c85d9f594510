#!/bin/bash
for m in small deep repeats contigparams longreads nrich verylong tiling tiny palrepeats letters uniform; do
  timeout -k 10 400 python scripts/fuzz_chain.py $m 200 9901 2>&1 | tail -n 1
done
