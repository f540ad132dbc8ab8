cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2p; mkdir -p $O
cd $R
python tools/partial_probe.py --nx 1000 --ny 1000 --k 100 --reps 5 --arms ";18=2;18=3" > $O/probe.jsonl 2>&1
LZ_NO_VMM=1 python tools/partial_probe.py --nx 1000 --ny 1000 --k 100 --reps 5 > $O/probe_novmm.jsonl 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 $R/tools/partial_probe.py --nx 1000 --ny 1000 --k 100 --reps 5 --no-profile > $O/probe_rocprof.jsonl 2>&1)
python3 tools/rocpd_stats.py $O/prof > $O/stats.txt 2>&1
rm -rf $O/prof
cat $O/probe.jsonl $O/probe_novmm.jsonl $O/probe_rocprof.jsonl | cut -c1-420; head -14 $O/stats.txt | cut -c1-150
