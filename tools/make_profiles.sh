# Regenerates the round's committed evidence under gpurun_out/ (copied to profiles/ by hand): bench line as the driver runs it, rocprofv3
# kernel stats of the same workload, the emulated-rank table, the config-5 rank step.
cd /root/repo
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench64_spectral.json 2> gpurun_out/r03_bench64.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu > /root/repo/gpurun_out/r03_bench64_spectral_under_rocprof.json 2>/dev/null; cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03_bench64_spectral_kernel_stats.csv)
python tools/emulate_rank.py --steps 5 > gpurun_out/r03_emulated_ranks.json 2> gpurun_out/r03_emulated_ranks.err
python tests/dryrun_config5.py --size 128 --world 8 --rank 0 > gpurun_out/r03_config5_rank0_of_8.json 2> gpurun_out/r03_config5.err
python bench.py --size 32 --kernel exp --drill 0 --steps 10 --warmup 2 --no-cpu > gpurun_out/r03_bench32_config2.json 2>/dev/null
