"""Run the rocprofv3 PMC passes (one --pmc group per run, kernel-trace/stats never combined with them) for ONE launch of a
kernel driver script and write the summary JSON that bench.py reads for roofline.traffic.

    python tools/pmc_collect.py posterior   -> gpurun_out/pmc_posterior_reduce.json   (driver: tools/run_posterior_once.py)

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KiB, and on gfx950
FETCH_SIZE counts 64 B per 128-B request, i.e. half of the bytes fetched (x2 correction)."""
import csv, glob, json, os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GROUPS = [["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY"],
          ["FETCH_SIZE"], ["WRITE_SIZE"],     # FETCH_SIZE costs 3 of the 4 TCC slots: one pass each
          ["TCC_HIT_sum", "TCC_MISS_sum"]]
# second argument "valu" (round 5): issue-side counters of the VALU-bound stream kernels -- where do the SIMD
# cycles that are not FMA issue go?  (SQ_INST_CYCLES_VALU does not exist on gfx9; SQ_ACTIVE_INST_* count in quad-cycles per SIMD.)
VALU_GROUPS = [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_THREAD_CYCLES_VALU"],
               ["SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"],
               ["SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_SALU"],
               ["SQ_WAVES", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_ACTIVE_INST_SCA"]]
DRIVERS = {"posterior": ("run_posterior_once.py", "gemm_f64_kernel<4, 2, 2, 1", "pmc_posterior_reduce.json"),
           "fused": ("run_fused_once.py", "gemm_f64_kernel<4, 2, 3", "pmc_ak_fused_grid.json"),
           "xz2d_fwd": ("run_spectral_kernels_once.py xz2d_fwd", "xz2d_kernel<64, 64, 128, 128>", "pmc_xz2d_fwd.json"),
           "xz2d_bwd": ("run_spectral_kernels_once.py xz2d_bwd", "xz2d_kernel<128, 128, 64, 64>", "pmc_xz2d_bwd.json"),
           "toeplitz": ("run_spectral_kernels_once.py toeplitz", "toeplitz_y_kernel", "pmc_toeplitz_y.json"),
           "toeplitz2t": ("run_spectral_kernels_once.py toeplitz2t", "toeplitz_y2_kernel", "pmc_toeplitz_y2t.json"),
           "toeplitz2s": ("run_spectral_kernels_once.py toeplitz2s", "toeplitz_y2s_kernel", "pmc_toeplitz_y2s.json"),
           "spectral_y": ("run_spectral_kernels_once.py spectral_y", "spectral_y_kernel<64, 1, 2", "pmc_spectral_y.json"),
           "spectral_y1": ("run_spectral_kernels_once.py spectral_y1", "spectral_y_kernel<64, 1, 1", "pmc_spectral_y1.json"),
           "spectral_y2s": ("run_spectral_kernels_once.py spectral_y2s", "spectral_y_kernel<64, 2, 2", "pmc_spectral_y2s.json"),
           "spectral_y128": ("run_spectral_kernels_once.py spectral_y128", "spectral_y_pipe_kernel<128, 1, 3", "pmc_spectral_y128.json"),
           "spectral_y128_1": ("run_spectral_kernels_once.py spectral_y128_1", "spectral_y_pipe_kernel<128, 1, 1", "pmc_spectral_y128_1.json"),
           "axis128_fwd": ("run_spectral_kernels_once.py axis128_fwd", "spectral_axis_kernel<128, false", "pmc_spectral_axis128_fwd.json"),
           "axis128_inv": ("run_spectral_kernels_once.py axis128_inv", "spectral_axis_kernel<128, true", "pmc_spectral_axis128_inv.json"),
           "xcorr": ("run_spectral_kernels_once.py xcorr", "xcorr_kernel", "pmc_xcorr.json"),
           "fold_fwd": ("run_spectral_kernels_once.py fold_fwd", "xz_fold_fwd_kernel", "pmc_xz2d_fold_fwd.json"),
           "fold_bwd": ("run_spectral_kernels_once.py fold_bwd", "xz_fold_inv_kernel", "pmc_xz2d_fold_bwd.json"),
           "xcorr_fold": ("run_spectral_kernels_once.py xcorr_fold", "xcorr_fold4_kernel", "pmc_xcorr_fold.json"),
           "xz2d32_fwd": ("run_spectral_kernels_once.py xz2d32_fwd", "xz2d_kernel<64, 32, 128, 64>", "pmc_xz2d32_fwd.json"),
           "xz2d32_bwd": ("run_spectral_kernels_once.py xz2d32_bwd", "xz2d_kernel<128, 64, 64, 32>", "pmc_xz2d32_bwd.json"),
           "ymul": ("run_spectral_kernels_once.py ymul", "ymul_kernel", "pmc_ymul.json"),
           "ymul_gemm": ("run_spectral_kernels_once.py ymul_gemm", "gemm_f64_kernel<2, 2, 2, 0", "pmc_ymul_as_gemm_batched.json"),
           "potrf_dag": ("run_potrf_once.py 8448 noctx", "potrf_dag_kernel", "pmc_potrf_dag.json"),
           "rank_update": ("run_rank_update_once.py", "gemm_f64_kernel<4, 2, 1, 0", "pmc_rank_update_k128.json"),
           "kblock_exp": ("run_k_block_once.py exp f64", "k_block_kernel", "pmc_k_block_exp_f64.json"),
           "kblock_matern": ("run_k_block_once.py matern32 f64", "k_block_kernel", "pmc_k_block_matern32_f64.json"),
           "kblock_exp_f32": ("run_k_block_once.py exp f32", "k_block_kernel", "pmc_k_block_exp_f32.json"),
           "toeplitz128": ("run_spectral_kernels_once.py toeplitz128", "toeplitz_y_win_kernel", "pmc_toeplitz_y_win128.json"),
           "kblock_grid": ("run_k_block_once.py matern32_x f64 grid", "k_block_grid_kernel", "pmc_k_block_grid_f64.json"),
           "fold_inv_ss": ("run_spectral_kernels_once.py fold_inv_ss", "xz_fold_inv_kernel<64, 1,", "pmc_xz2d_fold_inv_ss.json"),
           "fold_inv_strided": ("run_spectral_kernels_once.py fold_inv_strided", "xz_fold_inv_kernel<64, 0,", "pmc_xz2d_fold_inv_strided.json"),
           "fold_inv_mul": ("run_spectral_kernels_once.py fold_inv_mul", "xz_fold_inv_kernel<64, 2,", "pmc_xz2d_fold_inv_mul.json"),
           "wplanes": ("run_spectral_kernels_once.py wplanes", "lattice_wplanes_kernel", "pmc_lattice_wplanes.json"),
           "colgemv": ("run_spectral_kernels_once.py colgemv", "colgemv_kernel", "pmc_colgemv.json"),
           "kblock_grid_f32": ("run_k_block_once.py matern32_x f32 grid", "k_block_grid_kernel", "pmc_k_block_grid_f32.json")}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "posterior"
    script, kmatch, outname = DRIVERS[which]
    counters, text = {}, ""
    valu = len(sys.argv) > 2 and sys.argv[2] == "valu"
    if valu:
        outname = outname.replace(".json", "_valu.json")
    for grp in (GROUPS[:1] + VALU_GROUPS if valu else GROUPS):
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        sc = script.split()
        cmd = ["rocprofv3", "--pmc", *grp, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(HERE, sc[0]), *sc[1:]]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        text = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else text
        ndisp = {}
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if kmatch in row["Kernel_Name"]:
                    counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    ndisp.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
        for c, ids in ndisp.items():          # drivers that launch the kernel twice (warm-up + timed): per-launch average
            counters[c] /= len(ids)
    m = re.search(r"([0-9.]+) s, ([0-9.]+) T[FB]/s", text)
    secs = float(m.group(1)) if m else None
    mb = re.search(r"algorithmic bytes ([0-9][0-9.e+]*)", text)
    mf = re.search(r"flop ([0-9]+)", text)
    out = {"flop": float(mf.group(1)) if mf else (float(m.group(2)) * 1e12 * secs if m else None),"note": "rocprofv3 --pmc passes (separate runs, one launch each) of tools/%s: %s" % (script, text),
           "seconds_unprofiled_event": secs, "counters": counters, "derived": {}}
    d = out["derived"]
    if "GRBM_GUI_ACTIVE" in counters and secs:
        d["clock_GHz_during_profiled_pass"] = counters["GRBM_GUI_ACTIVE"] / 8.0 / secs / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in counters:
            d["mfma_busy_frac_of_simd_cycles"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / (counters["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    if valu and "GRBM_GUI_ACTIVE" in counters:
        simd_cycles = counters["GRBM_GUI_ACTIVE"] / 8.0 * 1024          # 8 XCDs report GUI_ACTIVE; 256 CUs x 4 SIMDs
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"):
            if k in counters:
                d[k.lower() + "_x4_over_simd_cycles"] = 4.0 * counters[k] / simd_cycles
        if counters.get("SQ_INSTS_VALU"):
            d["fma_f64_share_of_valu_instructions"] = counters.get("SQ_INSTS_VALU_FMA_F64", 0.0) / counters["SQ_INSTS_VALU"]
            d["valu_instructions_per_simd_cycle"] = counters["SQ_INSTS_VALU"] / simd_cycles
            d["fma_f64_issue_cycles_over_simd_cycles"] = 4.0 * counters.get("SQ_INSTS_VALU_FMA_F64", 0.0) / simd_cycles   # a wave64 fp64 FMA occupies its SIMD for 4 cycles... x2 on a half-rate pipe
        if counters.get("SQ_WAVES") and counters.get("SQ_WAIT_INST_LDS") is not None:
            d["wait_inst_lds_cycles_per_wave_x4"] = 4.0 * counters["SQ_WAIT_INST_LDS"] / counters["SQ_WAVES"]
        if counters.get("SQ_WAIT_ANY") is not None and counters.get("SQ_WAVES"):
            d["wait_any_cycles_per_wave_x4"] = 4.0 * counters["SQ_WAIT_ANY"] / counters["SQ_WAVES"]
    if counters.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = counters.get("SQ_LDS_BANK_CONFLICT", 0.0) / counters["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in counters:
        d["FETCH_GB_x2_gfx950_correction"] = counters["FETCH_SIZE"] * 1024 * 2 / 1e9
        d["WRITE_SIZE_GB"] = counters.get("WRITE_SIZE", 0.0) * 1024 / 1e9
        d["hbm_bytes_per_launch_corrected"] = counters["FETCH_SIZE"] * 1024 * 2 + counters.get("WRITE_SIZE", 0.0) * 1024
        if mb and secs:     # HBM-bound kernels: achieved bandwidth against the 8 TB/s roof, measured bytes against the algorithmic ones
            d["algorithmic_bytes"] = float(mb.group(1))
            d["algorithmic_TBps_unprofiled_event"] = float(mb.group(1)) / secs / 1e12
            d["measured_over_algorithmic_bytes"] = d["hbm_bytes_per_launch_corrected"] / float(mb.group(1))
            d["frac_of_8TBps_roof"] = float(mb.group(1)) / secs / 8e12
    if counters.get("TCC_HIT_sum") is not None and counters.get("TCC_MISS_sum"):
        d["L2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", outname), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
