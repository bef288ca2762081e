B=dasp_amd/bin/dasp_bench
for t in 16 32 64 128; do for w in HV15R nlpkkt160; do echo "== threads max $t $w: $(DASP_HOST_THREADS_MAX=$t $B $w 1 64 5 2 2>&1 | tail -1 | sed 's/.*fill0=[0-9.]* //; s/ win=.*//')"; done; done
