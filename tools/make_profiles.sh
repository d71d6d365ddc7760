# Regenerates the round's committed evidence under gpurun_out/ (copied to profiles/ by hand): bench line as the driver runs it, rocprofv3
# kernel stats of the same command, PMC summaries (HBM traffic of the y-stage kernels; issue-side counters of the VALU-bound ones; MFMA busy
# of the tile-DAG factorisation), the Cholesky A/B and its task timeline, the emulated-rank table, the one-rank RCCL check, the config-5
# rank step in the row form, config 2, the one-rank benches at 96^3 and 128^3 x 3, the stream probe.
# Run on the GPU box from the repository root:  bash tools/make_profiles.sh r06
R=${1:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$(pwd)
# y stage: the matrix-pipe kernels (round 6) and the direct kernels they replace -- HBM traffic, issue-side counters, lone-launch times
for k in spectral_y spectral_y2s spectral_y128 toeplitz toeplitz2s; do python tools/pmc_collect.py $k > /dev/null 2>&1; python tools/pmc_collect.py $k valu > /dev/null 2>&1; done
for f in pmc_spectral_y pmc_spectral_y2s pmc_spectral_y128 pmc_toeplitz_y pmc_toeplitz_y2s; do cp gpurun_out/$f.json gpurun_out/${R}_$f.json; cp gpurun_out/${f}_valu.json gpurun_out/${R}_${f}_valu.json; done
for k in fold_inv_mul fold_inv_ss; do python tools/pmc_collect.py $k valu > /dev/null 2>&1; done
cp gpurun_out/pmc_xz2d_fold_inv_mul_valu.json gpurun_out/${R}_pmc_xz2d_fold_inv_mul_valu.json; cp gpurun_out/pmc_xz2d_fold_inv_ss_valu.json gpurun_out/${R}_pmc_xz2d_fold_inv_ss_valu.json
python tools/pmc_collect.py kblock_grid > /dev/null 2>&1; cp gpurun_out/pmc_k_block_grid_f64.json gpurun_out/${R}_pmc_k_block_grid_f64.json
for k in potrf_dag axis128_fwd axis128_inv; do python tools/pmc_collect.py $k > /dev/null 2>&1; done
cp gpurun_out/pmc_potrf_dag.json gpurun_out/${R}_pmc_potrf_dag.json; cp gpurun_out/pmc_spectral_axis128_fwd.json gpurun_out/${R}_pmc_spectral_axis128_fwd.json; cp gpurun_out/pmc_spectral_axis128_inv.json gpurun_out/${R}_pmc_spectral_axis128_inv.json
for k in fold_fwd fold_bwd fold_inv_ss fold_inv_mul xcorr_fold ymul toeplitz toeplitz2s spectral_y spectral_y1 spectral_y2s spectral_y128 spectral_y3t128 axis128_fwd axis128_inv; do python tools/run_spectral_kernels_once.py $k 2>&1 | grep -v amdgpu.ids | tail -1; done > gpurun_out/${R}_spectral_kernels_once.txt
(python tools/time_toeplitz.py; python tools/time_toeplitz.py 96; python tools/time_toeplitz.py 128) 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_time_y_stage.txt
(python tools/time_axis_passes.py 96 16; python tools/time_axis_passes.py 128 16) 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_time_axis_passes.txt
# the driver's command, then the same under rocprofv3 (kernel stats), then the A/B with the direct y stage on the same box
python bench.py --steps 20 --warmup 5 2> gpurun_out/${R}_bench64.err | grep '^{' > gpurun_out/${R}_bench64_spectral.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > $ROOT/gpurun_out/${R}_bench64_spectral_under_rocprof.json; cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_bench64_spectral_kernel_stats.csv)
GEOBO_Y_MFMA=0 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench64_direct_y_same_box.json
# north_star's literal design on HEAD: the dense route (fused on-the-fly-K x A on MFMA), one step + its kernel stats
python bench.py --method dense --steps 1 --warmup 1 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench64_dense.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_d && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -- python $ROOT/bench.py --method dense --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1; cp $(find /tmp/prof_d -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_bench64_dense_kernel_stats.csv)
python bench.py --gpus 1 --backend nccl --check --steps 5 --warmup 2 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench64_one_rank_nccl_check.json
python bench.py --size 32 --kernel exp --drill 0 --steps 20 --warmup 3 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench32_config2.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python $ROOT/tests/dryrun_config5.py --size 128 --world 8 --rank 0 --no-oracle > $ROOT/gpurun_out/${R}_config5_rank0_under_rocprof.json 2>/dev/null; cp $(find /tmp/prof_c -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_config5_rank0_kernel_stats.csv)
python tools/emulate_rank.py --of 2,4,8 > gpurun_out/${R}_emulated_ranks.json 2>/dev/null
python tests/dryrun_config5.py --size 128 --world 8 --rank 0 --repeat 2 > gpurun_out/${R}_config5_rank0_of_8_rowform.json 2> gpurun_out/${R}_config5_rank0.err
python bench.py --size 96 --steps 2 --warmup 1 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench96_one_gpu.json
python bench.py --size 128 --props 3 --assembly f32 --steps 1 --warmup 1 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/${R}_bench128x3_f32_one_gpu.json
GEOBO_HIP_LIB= python tools/run_potrf_once.py 2048 noctx 2>&1 | tail -1 > gpurun_out/${R}_potrf_times.txt; python tools/run_potrf_once.py 8448 noctx 2>&1 | tail -1 >> gpurun_out/${R}_potrf_times.txt
python tools/soak_kernels.py 41 180 > gpurun_out/${R}_soak_final_kernels.txt 2>&1
