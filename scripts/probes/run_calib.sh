#!/bin/bash
# one gpurun call: the keys-only radix probe and the PMC calibration (plain, FETCH_SIZE pass, WRITE_SIZE pass)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 300 $R/scripts/probes/rx8_bench.bin > $O/r05_rx8_bench.txt 2>&1; echo "rx8 rc=$?"; cat $O/r05_rx8_bench.txt
timeout -k 10 200 $R/scripts/probes/pmc_calib.bin > $O/r05_pmc_calib_plain.txt 2>&1 || exit 1
cat $O/r05_pmc_calib_plain.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calf -o p --output-format csv -- $R/scripts/probes/pmc_calib.bin > $O/r05_pmc_calib_f.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calw -o p --output-format csv -- $R/scripts/probes/pmc_calib.bin > $O/r05_pmc_calib_w.txt 2>&1 || exit 1
cd $R
python scripts/probes/pmc_calib.py $O/r05_pmc_calib_plain.txt $O/calf/p_counter_collection.csv $O/calw/p_counter_collection.csv $O/r05_pmc_calibration.json
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0?_RDREQ|TCC_EA0?_WRREQ|TCC_BUBBLE|TCC_EA0_RD" | head -20 > $O/r05_tcc_counters.txt; cat $O/r05_tcc_counters.txt
