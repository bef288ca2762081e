#!/bin/bash
# tools/traffic_all.sh -- the three rocprofv3 passes (kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE: separate runs) of
# dasp_bench for every workload bench.py reports; run through gpurun from the repo root, then tools/traffic.py turns each
# gpurun_out/prof_<tag>/ into a profiles/traffic.json entry.
B=dasp_amd/bin/dasp_bench
while read tag w sc pr it; do
  tools/prof.sh $tag -- $B $w $sc $pr $it 3 > /dev/null 2>&1
  tail -1 gpurun_out/prof_$tag/trace.log | cut -c1-200
done <<LIST
hv15r64 HV15R 1 64 20
cop64 cop20k_A 1 64 200
nlp64 nlpkkt160 1 64 20
pl64 powerlaw_1M 1 64 20
queen64 Queen_4147 1 64 20
wb16 webbase-1M 1 16 200
lj16 ljournal-2008 1 16 20
rmat16 rmat_2M 1 16 50
lju16 ljournal-2008-uniform 1 16 20
hvu64 HV15R-unstructured 1 64 20
wbu16 webbase-1M-uniform 1 16 200
LIST
