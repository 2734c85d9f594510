"""Stage flags exactly as `carpedeam ancient_assemble` emits them on defaults."""
K_FLAGS = ("--kmer-per-seq 200 --kmer-per-seq-scale 0.2 --hash-shift 67 --ignore-multi-kmer 1 --mask 0 "
           "--adjust-kmer-len 0 --cov-mode 1 -c 0 --include-only-extendable 0 -k 20").split()
R_FLAGS = ("--rescore-mode 3 -e 0.001 --min-seq-id 0.9 --seq-id-mode 0 --sort-results 0 -a 0 --filter-hits 0 "
           "--cov-mode 1 -c 0").split()
A_FLAGS = ("--rescore-mode 3 --max-seq-len 200000 --min-seq-id 0.9 --ext-random-align 0.85 --excess-penalty 0.0625 "
           "--min-ryseq-id-corr-reads 0.99 --likelihood-ratio-threshold 0.5 --unsafe 0 --min-cov-safe 5").split()
# the contig phase (data/nuclassemble.sh:148-196; Nuclassembler.cpp:118-126): -k 22, only extendable overlaps, --min-merge-seq-id
KC_FLAGS = [x if x != "20" else "22" for x in K_FLAGS]
KC_FLAGS[KC_FLAGS.index("--include-only-extendable") + 1] = "1"
AC_FLAGS = A_FLAGS + ["--min-merge-seq-id", "0.99"]

# linclust's pre-clustering of the assembled contigs as `ancient_assemble` runs it (lib/mmseqs/data/workflow/linclust.sh:21-31,
# src/workflow/GuidedNuclassembler.cpp:176-181): its kmermatcher and its Hamming-distance rescorediagonal
LINCLUST_K_FLAGS = ("--alph-size nucl:5,aa:13 --min-seq-id 0.97 --kmer-per-seq 200 --spaced-kmer-mode 0 --kmer-per-seq-scale 0.200 --adjust-kmer-len 0 --mask 0 "
                    "--mask-lower-case 0 --cov-mode 1 -k 20 -c 0.99 --max-seq-len 200000 --hash-shift 67 --split-memory-limit 0 --include-only-extendable 0 "
                    "--ignore-multi-kmer 1").split()
HAMMING_FLAGS = ("--rescore-mode 0 --wrapped-scoring 1 --filter-hits 0 -e 0.001 -c 0.99 -a 0 --cov-mode 1 --min-seq-id 0.97 --min-aln-len 0 --seq-id-mode 0 "
                 "--add-self-matches 0 --sort-results 0").split()
