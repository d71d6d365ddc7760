cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python /root/repo/tests/dryrun_config5.py --size 128 --world 8 --rank 0 --no-oracle > /root/repo/gpurun_out/c5_prof.json 2> /root/repo/gpurun_out/c5_prof.err
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
cp $f /root/repo/gpurun_out/c5_kernel_stats.csv
