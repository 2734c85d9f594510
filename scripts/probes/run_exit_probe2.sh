#!/bin/bash
# is memory a process RELEASES cleared while that process lives on (so that its successor finds it clean)?
P=scripts/probes/exit_probe.bin
$P 100 256 leave > /dev/null; sleep 4
echo "== A releases 100 GB and stays alive 3 s; B follows at once"; $P 100 256 release vmm 3; $P 100 256 leave
sleep 4
echo "== A releases 100 GB and leaves at once; B follows at once"; $P 100 256 release vmm 0; $P 100 256 leave
sleep 4
echo "== A leaves with 100 GB mapped; B (10 GB) follows at once, then C (10 GB)"; $P 100 256 leave; $P 10 256 leave; $P 10 256 leave
