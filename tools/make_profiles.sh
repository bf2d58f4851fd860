#!/bin/bash
# Regenerates what profiles/ holds for the current tree, on a GPU box (run from the repository root):
#   bash tools/make_profiles.sh <tag>      e.g. r3a
# writes gpurun_out/<tag>_bench_n512_kernel_stats.csv, gpurun_out/<tag>_step_kernels.txt, gpurun_out/<tag>_host_vs_device.txt,
# gpurun_out/pmc_traffic.json and
# gpurun_out/pmc_mfma.json (copy them into profiles/; the two JSON files take the commit from csrc/BUILD_SHA).
# Every rocprofv3 pass is a run of its own; counters are never combined with trace options beyond --kernel-trace.
set -e
tag=${1:-r3}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --soak 0"
rm -rf /tmp/prof_trace /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o x -- $cmd > $out/${tag}_trace.log 2>&1
cp $(find /tmp/prof_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_n512_kernel_stats.csv
echo "trace done"
# The kernel listing of one minibatch comes from a second trace in which the device idles 5 ms in front of every
# minibatch (ICRL_BENCH_HEAD_START): under the profiler every launch costs the host about 10 us, the host falls behind
# the device, and launches then start in the order the host gets to them - not the schedule an unprofiled run has,
# where the host is 0.8 ms ahead of the device when a minibatch starts (tools/pipe_timeline.py, below).
rm -rf /tmp/prof_list
ICRL_BENCH_HEAD_START=12000000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_list -o x -- $cmd > $out/${tag}_list.log 2>&1
python3 $root/tools/step_kernels.py /tmp/prof_list 6 20 > $out/${tag}_step_kernels.txt
echo "listing done"
cd $root
timeout -k 10 200 python3 tools/pipe_timeline.py 8 2 > $out/${tag}_host_vs_device.txt 2> $out/${tag}_timeline.log
echo "--- after 170 minibatches (the warm-ups have grown) ---" >> $out/${tag}_host_vs_device.txt
timeout -k 10 300 python3 tools/pipe_timeline.py 170 1 >> $out/${tag}_host_vs_device.txt 2>> $out/${tag}_timeline.log
cd /tmp
echo "timeline done"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- $cmd > $out/${tag}_pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- $cmd > $out/${tag}_pmc_write.log 2>&1
echo "write done"
python3 $root/tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_traffic.json > $out/${tag}_pmc_traffic.txt
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- $cmd > $out/${tag}_pmc_mfma.log 2>&1
echo "mfma done"
python3 $root/tools/pmc_mfma_summary.py /tmp/pmc_mfma $out/pmc_mfma.json > $out/${tag}_pmc_mfma.txt
sha=$(cat $root/image-captioning-through-rl_amd/csrc/BUILD_SHA 2>/dev/null || echo unknown)
python3 - "$sha" $out/pmc_traffic.json $out/pmc_mfma.json <<'PY'
import json, sys
for p in sys.argv[2:]:
    d = json.load(open(p))
    if not d.get("_git_sha"):
        d["_git_sha"] = sys.argv[1].strip()
    json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY
echo "profiles written for $sha"
