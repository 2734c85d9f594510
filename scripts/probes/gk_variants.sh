#!/bin/bash
# k_bucket_groups under variants of the library (rocprofv3 kernel statistics of kmermatcher alone on 50 M reads)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-default packed}; do
  unset CDM_LIB CDM_KMER_LAYOUT
  case $v in default) ;; packed) export CDM_KMER_LAYOUT=packed;; *) export CDM_LIB=$R/carpedeam_amd/_variants/libcarpedeam_hip_$v.so;; esac
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/gkv_$v -o p --output-format csv -- python3 $R/scripts/km_only.py 50000000 3 > $O/gkv_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/gkv_$v.log; continue; }
  echo "== $v: $(python3 $R/scripts/kstats.py $(find $O/gkv_$v -name p_kernel_stats.csv) 3 12 | head -${TOPN:-3})"; grep kmermatch $O/gkv_$v.log | tail -1 | cut -c1-150
  rm -rf $O/gkv_$v
done
