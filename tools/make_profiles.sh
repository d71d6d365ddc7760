# Regenerates the round's committed evidence under gpurun_out/ (copied to profiles/ by hand): bench line as the driver runs it, rocprofv3
# kernel stats of the same command, the emulated-rank table, the config-5 rank step in the row form, config 2, the one-rank benches at
# 96^3 and 128^3 x 3, the Cholesky A/B.  Run on the GPU box from the repository root:  bash tools/make_profiles.sh r05
R=${1:-r05}
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$(pwd)
python tools/pmc_collect.py toeplitz2t > /dev/null 2>&1; cp gpurun_out/pmc_toeplitz_y2t.json gpurun_out/${R}_pmc_toeplitz_y2t.json; python tools/pmc_collect.py toeplitz > /dev/null 2>&1; cp gpurun_out/pmc_toeplitz_y.json gpurun_out/${R}_pmc_toeplitz_y.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench64_spectral.json 2> gpurun_out/${R}_bench64.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu > $ROOT/gpurun_out/${R}_bench64_spectral_under_rocprof.json 2>/dev/null; cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_bench64_spectral_kernel_stats.csv)
python bench.py --size 32 --kernel exp --drill 0 --steps 20 --warmup 3 --no-cpu > gpurun_out/${R}_bench32_config2.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python $ROOT/tests/dryrun_config5.py --size 128 --world 8 --rank 0 --no-oracle > $ROOT/gpurun_out/${R}_config5_rank0_under_rocprof.json 2>/dev/null; cp $(find /tmp/prof_c -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_config5_rank0_kernel_stats.csv)
python tools/check_potrf_dag.py 1024 2048 4224 8448 16640 33024 > gpurun_out/${R}_potrf_dag_vs_streams.txt 2>&1
python tools/pmc_collect.py potrf_dag > /dev/null 2>&1; cp gpurun_out/pmc_potrf_dag.json gpurun_out/${R}_pmc_potrf_dag.json
# (task timeline of the tile DAG: python -c "from geobo_amd.build import build; build(extra_flags=('-DGEOBO_DAG_TRACE',))"; python tools/potrf_dag_trace.py 8448 gpurun_out/${R}_potrf_dag_trace_8448.txt; python -m geobo_amd.build)
python tests/dryrun_config5.py --size 128 --world 8 --rank 0 > gpurun_out/${R}_config5_rank0_of_8_rowform.json 2> gpurun_out/${R}_config5_rank0.err
python bench.py --size 96 --steps 2 --warmup 1 --no-cpu > gpurun_out/${R}_bench96_one_gpu.json 2>/dev/null
python bench.py --size 128 --props 3 --assembly f32 --steps 1 --warmup 1 --no-cpu > gpurun_out/${R}_bench128x3_f32_one_gpu.json 2>/dev/null
