#!/bin/bash
# pairs of processes back to back: what the second one pays after the first one left / released its memory
P=scripts/probes/exit_probe.bin
t() { local s=$(date +%s.%N); "$@"; local e=$(date +%s.%N); echo "   wall $(echo "$e - $s" | bc -l | cut -c1-6) s"; }
for mode in leave release; do for chunk in 256 2048; do
  echo "== $mode, chunks of $chunk MB"; t $P 140 $chunk $mode; t $P 140 $chunk $mode; sleep 3; t $P 140 $chunk $mode
done; done
echo "== hipMalloc, leave"; t $P 140 2048 leave malloc; t $P 140 2048 leave malloc
