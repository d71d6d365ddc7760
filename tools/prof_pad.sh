cd /tmp && export TMPDIR=/tmp
for PAD in 256 0; do
  export GEOBO_SPECTRAL_PLANE_PAD=$PAD
  rm -rf /tmp/prof_$PAD
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$PAD -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu > /root/repo/gpurun_out/bench_pad$PAD.json 2>/dev/null
  f=$(find /tmp/prof_$PAD -name "*kernel_stats.csv" | head -1)
  cp $f /root/repo/gpurun_out/kernel_stats_pad$PAD.csv
done
cd /root/repo
python -m pytest tests/test_inversion_gpu.py -m gpu -x -q -k "cube16 or tiny or cube32" 2>&1 | tail -3
