B=dasp_amd/bin/dasp_bench
DASP_VERBOSE=1 $B HV15R 1 64 20 5 2>&1 | grep -v "^$" | tail -21
for w in nlpkkt160 Queen_4147 cop20k_A ljournal-2008; do $B $w 1 64 20 5 2>&1 | tail -1 | sed 's/fill0.*pre=/pre=/; s/win=.*//'; done
