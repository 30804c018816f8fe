#!/bin/bash
# The end-of-round measurement campaign in one call, on the GPU box:   bash tools/campaign.sh <tag>     (e.g. r3v14)
# GPU suite, fuzz, frame hashes; the PMC passes and profiles/traffic.json (written by tools/pmc_summary.py only); bench.py;
# the serialised run under rocprofv3 --kernel-trace --stats; one rank's share for N = 1, 2, 4, 8; the configs table; launch
# costs.  Every GPU step under its own time limit, the chain stops at the first failure.  Results under gpurun_out/<tag>/
# (copy what is to be judged into profiles/r0N/).
tag=${1:?tag}
root=$(cd "$(dirname "$0")/.." && pwd)
O=$root/gpurun_out/$tag
mkdir -p $O
cd $root
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
grep -q " passed" $O/pytest_gpu.log && ! grep -q failed $O/pytest_gpu.log || exit 1
timeout -k 10 400 python3 -u tools/fuzz_parity.py 300000 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
grep -q "TOTAL MISMATCHES 0" $O/fuzz.log || exit 1
timeout -k 10 100 python3 tools/image_hash.py 2>&1 | tail -4 | tee $O/image_hash.txt || exit 1
bash tools/pmc_passes.sh $tag || exit 1
python3 tools/pmc_summary.py $tag r03 --traffic > $O/pmc_summary.log 2>&1 || exit 1
cp profiles/r03/pmc_$tag.json profiles/traffic.json $O/
timeout -k 10 300 python3 bench.py > $O/bench_$tag.json 2> $O/bench.err || exit 1
python3 tools/print_bench.py $O/bench_$tag.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$tag && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap > $O/bench_${tag}_serialised_under_rocprof.json 2> $O/rocprof.err && cp $(find /tmp/ks_$tag -name "*kernel_stats.csv") $O/bench_${tag}_kernel_stats.csv ) || exit 1
: > $O/rehearse_rank_$tag.txt
for p in 1 2 4 8; do timeout -k 10 120 python3 tools/rehearse_rank.py --strong --parts $p --part $((p/2)) --steps 5 >> $O/rehearse_rank_$tag.txt || exit 1; done
timeout -k 10 120 python3 tools/rehearse_rank.py --parts 8 --part 4 --steps 3 >> $O/rehearse_rank_$tag.txt || exit 1
cat $O/rehearse_rank_$tag.txt
timeout -k 10 400 python3 tools/config_times.py > $O/config_times_$tag.jsonl 2> $O/config_times.err || exit 1
timeout -k 10 120 python3 tools/launch_linearity.py > $O/launch_linearity_$tag.txt || exit 1
timeout -k 10 120 python3 tools/small_launch.py > $O/small_launch_$tag.txt || exit 1
timeout -k 10 100 python3 tools/slow_rays_probe.py > $O/slow_rays_probe_$tag.txt || exit 1
echo "campaign $tag done"
