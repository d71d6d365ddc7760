# Regenerates the round's committed evidence under gpurun_out/ (copied to profiles/ by hand): bench line as the driver runs it, rocprofv3
# kernel stats of the same command, PMC summaries (HBM traffic of the y-stage kernels; issue-side counters of the VALU-bound ones; MFMA busy
# of the tile-DAG factorisation), the Cholesky A/B and its task timeline, the emulated-rank table, the one-rank RCCL check, the config-5
# rank step in the row form, config 2, the one-rank benches at 96^3 and 128^3 x 3, the stream probe.
# Run on the GPU box from the repository root:  bash tools/make_profiles.sh r05
R=${1:-r05}
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$(pwd)
python tools/pmc_collect.py toeplitz2s > /dev/null 2>&1; cp gpurun_out/pmc_toeplitz_y2s.json gpurun_out/${R}_pmc_toeplitz_y2s.json; python tools/pmc_collect.py toeplitz > /dev/null 2>&1; cp gpurun_out/pmc_toeplitz_y.json gpurun_out/${R}_pmc_toeplitz_y.json
for k in toeplitz toeplitz2s fold_inv_mul fold_inv_ss; do python tools/pmc_collect.py $k valu > /dev/null 2>&1; done
cp gpurun_out/pmc_toeplitz_y_valu.json gpurun_out/${R}_pmc_toeplitz_y_valu.json; cp gpurun_out/pmc_toeplitz_y2s_valu.json gpurun_out/${R}_pmc_toeplitz_y2s_valu.json
for k in fold_fwd fold_bwd fold_inv_ss fold_inv_mul xcorr_fold ymul toeplitz toeplitz2t toeplitz2s; do python tools/run_spectral_kernels_once.py $k 2>&1 | grep -v amdgpu.ids | tail -1; done > gpurun_out/${R}_spectral_kernels_once.txt
cp gpurun_out/pmc_xz2d_fold_inv_mul_valu.json gpurun_out/${R}_pmc_xz2d_fold_inv_mul_valu.json; cp gpurun_out/pmc_xz2d_fold_inv_ss_valu.json gpurun_out/${R}_pmc_xz2d_fold_inv_ss_valu.json
python bench.py --steps 20 --warmup 5 2> gpurun_out/${R}_bench64.err | grep '^{' > gpurun_out/${R}_bench64_spectral.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > $ROOT/gpurun_out/${R}_bench64_spectral_under_rocprof.json; cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_bench64_spectral_kernel_stats.csv)
python bench.py --gpus 1 --backend nccl --check --steps 5 --warmup 2 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench64_one_rank_nccl_check.json
python bench.py --size 32 --kernel exp --drill 0 --steps 20 --warmup 3 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench32_config2.json
GEOBO_ROWS=1 python bench.py --size 32 --kernel exp --drill 0 --steps 20 --warmup 3 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench32_rowform_forced.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python $ROOT/tests/dryrun_config5.py --size 128 --world 8 --rank 0 --no-oracle > $ROOT/gpurun_out/${R}_config5_rank0_under_rocprof.json 2>/dev/null; cp $(find /tmp/prof_c -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_config5_rank0_kernel_stats.csv)
python tools/check_potrf_dag.py 1024 2048 4224 8448 16640 33024 2>&1 | grep "m=\|bit" > gpurun_out/${R}_potrf_dag_vs_streams.txt
python tools/pmc_collect.py potrf_dag > /dev/null 2>&1; cp gpurun_out/pmc_potrf_dag.json gpurun_out/${R}_pmc_potrf_dag.json
python tools/emulate_rank.py --of 2,4,8 > gpurun_out/${R}_emulated_ranks.json 2>/dev/null
python tests/dryrun_config5.py --size 128 --world 8 --rank 0 > gpurun_out/${R}_config5_rank0_of_8_rowform.json 2> gpurun_out/${R}_config5_rank0.err
python bench.py --size 96 --steps 2 --warmup 1 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench96_one_gpu.json
python bench.py --size 128 --props 3 --assembly f32 --steps 1 --warmup 1 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench128x3_f32_one_gpu.json
[ -x tools/bin/hbm_copy_runs ] && tools/bin/hbm_copy_runs 256 2>&1 | head -22 > gpurun_out/${R}_hbm_copy_runs.txt
# task timeline of the tile DAG (a traced build of the library, then the default one again)
python -c "from geobo_amd.build import build; build(extra_flags=('-DGEOBO_DAG_TRACE',))" > /dev/null 2>&1 && python tools/potrf_dag_trace.py 8448 gpurun_out/${R}_potrf_dag_trace_8448.txt > /dev/null 2>&1; python tools/potrf_dag_trace.py 2048 gpurun_out/${R}_potrf_dag_trace_2048.txt > /dev/null 2>&1; python -m geobo_amd.build > /dev/null 2>&1
