cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu > /root/repo/gpurun_out/bench_prof.json 2>/dev/null
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/bench_prof_kernel_stats.csv
cp $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) /tmp/kt.csv
python /root/repo/tools/step_timeline.py /tmp/kt.csv > /root/repo/gpurun_out/bench_prof_timeline.txt
