cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_em
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_em -- python /root/repo/tools/emulate_rank.py --of 8 --steps 2 --warmup 1 > /root/repo/gpurun_out/emul8_prof.json 2> /root/repo/gpurun_out/emul8_prof.err
f=$(find /tmp/prof_em -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/step_timeline.py $f > /root/repo/gpurun_out/emul8_timeline.txt
