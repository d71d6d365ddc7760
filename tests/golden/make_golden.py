#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (sebhaan/geobo).

This script only works in the build container, where the reference checkout is mounted
read-only at /root/reference.  It is committed so that every fixture in this directory can be
traced to the reference call that produced it; the reference's Python never travels to the GPU
box -- only the numeric inputs/outputs written here do.

How the reference is imported:
  * geobo/kernels.py:23 does `from scipy import reshape, sqrt, identity` (aliases removed from
    modern SciPy and never used) -> we pre-set those names on the scipy module;
  * geobo/config_loader.py:20-39 reads sys.argv[1] at import time and freezes the YAML keys into
    module globals -> one *subprocess per settings file*;
  * run_geobo.py needs rasterio / pyvista / skimage which are not installed -> tiny stand-in
    modules that live only in a temp dir for the duration of this script (harness-side stubs of
    I/O packages, not of reference code).

Fixtures written (all float64, little endian, np.savez_compressed):
  F0  kat_kernels.npz          scalar / 3x3 known answers of geobo/kernels.py
  F1  tiny_<kernel>.npz        non-cubic 10x8x6 grid, kernels exp / sparse / matern32
  F2  cube16_<kernel>.npz      16^3 grid (exp with M_d=0, matern32 with 50 drill rows, sparse with 8)
  F3  example{1,2}.npz         cubing() inputs captured from run_geobo.py on the shipped examples,
                               its six output cubes, and the committed examples/results/*/*.vtk cubes
  F4  forward_kat.npz          simcube_cylinders.csv -> simsurveydata_cylinders.csv (A_sens known answer)
  F5  config1_exp16.npz        BASELINE config 1: examples/settings_example1.yaml extents (3050 x 1952 x 800 m -> anisotropic
                               190.6 x 122 x 50 m voxels) with xNcube = yNcube = zNcube = 16, kernelfunc 'exp',
                               gp_coeff = [0, 0, 0] ("gravity only": the blocks decouple), no drill rows
  F6  illcond_<name>.npz       long length scales / small noise / amplitude 2 (the regime optimize_gp explores): tiny grid with
                               the reference's operators, AkA and factor diagonal, and a 16^3 cube
  F7  optimize_tiny_exp.npz    Inversion.optimize_gp() (SHGO over 5 hyper-parameters) on the tiny grid: optimum, objective there,
                               and calc_logl at a few fixed parameter vectors

Usage:  python tests/golden/make_golden.py [F0 F1 F2 F3 F4 F5 F6 F7]
"""
import json
import os
import subprocess
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

BASE_SETTINGS = dict(
    inpath="/tmp/geobo_golden/in/", outpath="/tmp/geobo_golden/out/",
    FNAME_drilldata="none.csv", FNAME_gravsurvey="none.tif", FNAME_magsurvey="none.tif",
    drill_features=["DENSITY", "MAGSUS"], ifeature=0,
    xmin=0, xmax=1000, ymin=0, ymax=800, zmax=0, zoff=1, zLcube=600.0,
    xNcube=10, yNcube=8, zNcube=6,
    gen_simulation=False, modelname="cylinders",
    gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2],
    kernelfunc="exp", optimize_gp=False,
    XMAG=0, YMAG=0, ZMAG=1,
    plot_vertical=False, plot3d=False, regrid_sparse=False, font_scale=1.5,
    bayesopt_vertical=False, bayesopt_nonvertical=False, kappa=1, beta=0.0,
    c_G=6.673848e-11, c_SI_TO_MILLIGALS=10000, c_GCM3_TO_SI=1000.0, fcor_grav=1.0, fcor_mag=0.001,
)

# ----------------------------------------------------------------------------------------------
# worker: runs inside a fresh interpreter, one settings file per process
# ----------------------------------------------------------------------------------------------
WORKER = r'''
import sys, json, os, numpy as np, scipy, warnings
warnings.filterwarnings("ignore")
for n in ("reshape", "sqrt", "identity"):
    setattr(scipy, n, getattr(np, n))
job = json.load(open(sys.argv[1]))
sys.path.insert(0, job["stubdir"]); sys.path.insert(1, "/root/reference")
sys.argv = ["x", job["settings_yaml"]]
import geobo.config_loader as cfg
import geobo.kernels as K, geobo.sensormodel as sm, geobo.inversion as inv

def synthetic_inputs(I, md, chi_factor=None):
    """cylinders ground truth (simcube.py:83-92 semantics) -> survey through the reference A_sens,
    rounded through float32 like the GeoTIFF round trip (simcube.py:196-199)."""
    nx, ny, nz = cfg.xNcube, cfg.yNcube, cfg.zNcube
    x3, y3, z3 = I.xxx, I.yyy, I.zzz
    rad = cfg.yLcube / 18.
    rc1 = ((y3 - cfg.yLcube/1.3 - rad)**2) + ((z3 + cfg.zLcube/4 - rad)**2)
    rc2 = ((y3 - cfg.yLcube/4. - rad)**2) + ((z3 + cfg.zLcube/4 - rad)**2)
    rho = x3 * 0. + 0.1
    rho[rc2 <= rad**2] = 1.
    rho[rc1 <= rad**2] = 1.
    rho[(x3 < cfg.xLcube/5.) | (x3 > cfg.xLcube*4./5.)] = 0.1
    # smooth trend on top of the reference's cylinders model so that small grids (where no voxel
    # centre falls inside a cylinder) still give non-degenerate survey and drill data
    rho = rho + 0.02 * (x3 / cfg.xLcube + 2. * y3 / cfg.yLcube - z3 / cfg.zLcube)
    chi = (cfg.gp_coeff[1] if chi_factor is None else chi_factor) * rho   # (gp_coeff = 0 would give a constant-zero magnetic survey)
    xs = np.linspace(0.5, nx - 0.5, nx) * cfg.xvoxsize
    ys = np.linspace(0.5, ny - 0.5, ny) * cfg.yvoxsize
    X, Y, Z = np.meshgrid(xs, ys, cfg.zmax + cfg.zoff)
    loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
    Ag, _ = sm.A_sens(cfg.magneticField * 0., loc, I.Edges, 'grav')
    Am, _ = sm.A_sens(cfg.magneticField, loc, I.Edges, 'magn')
    grav = (Ag @ rho.flatten()).astype(np.float32).astype(np.float64)
    mag = (Am @ chi.flatten()).astype(np.float32).astype(np.float64)
    drill0 = np.zeros_like(rho)
    if md > 0:
        sel = np.random.default_rng(2020).choice(rho.size, md, replace=False)
        d = drill0.reshape(-1); d[sel] = rho.reshape(-1)[sel]
    return grav, mag, loc, drill0, rho, chi

out = {}
mode = job["mode"]
if mode == "kat":
    out["k2_0"] = K.gpkernel2(0., [600., 650.])
    out["m2_0"] = K.gpkernel_matern32_2(0., [600., 650.])
    pts = np.array([[0., 0., 0.], [100., 0., 0.], [100., 250., 75.]])
    D2 = K.calcDistanceMatrix(pts)
    out["D2"] = D2
    for name in ("exp", "sparse", "matern32"):
        out["cov_eq_" + name] = K.create_cov(D2, np.array([200., 200., 200.]), [1.0, 0.2, 0.2], fkernel=name)
        out["cov_ne_" + name] = K.create_cov(D2, np.array([200., 230., 270.]), [0.7, 0.3, 0.2], fkernel=name)
    gl = np.array([200., 200., 200.]); K.create_cov(D2, gl, [1, 1, 1], fkernel="exp"); out["mutated_eq"] = gl
    gl = np.array([200., 300., 200.]); K.create_cov(D2, gl, [1, 1, 1], fkernel="exp"); out["mutated_20"] = gl
    gl = np.array([200., 300., 300.]); K.create_cov(D2, gl, [1, 1, 1], fkernel="exp"); out["mutated_21"] = gl
    d2 = np.linspace(0., 1000.**2, 41)
    out["d2_line"] = d2
    out["k_exp"] = K.gpkernel(d2, 200.); out["k_exp2"] = K.gpkernel2(d2, [200., 204.])
    out["k_sp"] = K.gpkernel_sparse(d2, 400.); out["k_sp2"] = K.gpkernel_sparse2(d2, [400., 408.])
    out["k_sp2_eq"] = K.gpkernel_sparse2(d2, [400., 400.])
    out["k_m"] = K.gpkernel_matern32(d2, 200.); out["k_m2"] = K.gpkernel_matern32_2(d2, [200., 204.])
    out["points3D"] = K.calcGridPoints3D((3, 2, 4), (10., 20., 5.))
elif mode == "cubing":
    I = inv.Inversion(); vox = I.create_cubegeometry()
    grav, mag, loc, drill0, rho, chi = synthetic_inputs(I, job["md"], job.get("chi_factor"))
    if job.get("gp_length") is not None:
        I.gp_length = np.array(job["gp_length"], dtype=float)
    if job.get("gp_amp") is not None:
        I.gp_amp = float(job["gp_amp"])
    out["gp_amp"] = float(I.gp_amp)
    out["gp_length_in"] = np.array(I.gp_length, dtype=float)
    cubes = I.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    out.update(gravfield=grav, magfield=mag, sensor_locations=loc, drilldata0=drill0, rho=rho, chi=chi,
               voxelpos=vox, Edges=I.Edges, cubes=np.asarray(cubes), mu=I.mu_rec, var=np.diag(I.cov_rec),
               logl=I.logl, gp_length_out=np.array(I.gp_length, dtype=float), Fs3=I.Fs3,
               sel=np.flatnonzero(drill0.reshape(-1) != 0))
    ng = grav.size
    if job.get("save_A", False):
        out["A_g"] = I.Asens3[:ng, :rho.size]; out["A_m"] = I.Asens3[ng:2*ng, rho.size:2*rho.size]
    yerr = np.hstack((grav*0 + I.gp_sigma[0], mag*0 + I.gp_sigma[1], I.drillfield*0 + I.gp_sigma[2]))
    AkA = I.Asens3 @ (I.kcov @ I.Asens3.T) + np.diag(yerr**2)
    if job.get("save_AkA", False):
        out["AkA"] = AkA
    out["cond_AkA"] = np.linalg.cond(AkA)
    if job.get("save_Ldiag", False):
        import scipy.linalg as sl
        out["L_diag"] = np.diag(sl.cholesky(AkA, lower=True))
    if job.get("optimize", False):
        # Inversion.optimize_gp (inversion.py:155-178) on the state cubing() left behind.  NB the reference then stores the bare
        # scalar lengthscale in self.gp_length (:175), after which create_cov can no longer index it -- so it is called here,
        # after cubing, not through the YAML switch
        import io, contextlib
        probes = [[1.0, 2.0, 1.0, 0.2, 0.2], [1.3, 1.7, 0.8, 0.25, 0.3], [0.6, 6.0, 0.9, 0.5, 0.15], [2.0, 18.0, 0.6, 0.9, 0.9]]
        out["probe_params"] = np.asarray(probes)
        out["probe_values"] = np.asarray([I.calc_logl(np.asarray(pp)) for pp in probes])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            I.optimize_gp()
        out["opt_stdout"] = buf.getvalue()
        out["opt_gp_amp"] = float(I.gp_amp)
        out["opt_gp_length"] = np.asarray(I.gp_length, dtype=float)      # scalar multiple of the voxel size (sic)
        out["opt_coeffm"] = np.asarray(I.coeffm, dtype=float)
        xo = np.r_[float(I.gp_amp), float(np.asarray(I.gp_length).reshape(-1)[0]), np.asarray(I.coeffm, dtype=float)]
        out["opt_x"] = xo
        out["opt_fun"] = I.calc_logl(xo)
    # row-class statistics of the operators (used for the T2 operator-parity tier)
    out["A_g_absmax"] = np.abs(I.Asens3[:ng, :rho.size]).max()
    out["A_m_absmax"] = np.abs(I.Asens3[ng:2*ng, rho.size:2*rho.size]).max()
    out["A_g_rowsum"] = I.Asens3[:ng, :rho.size].sum(axis=1)
    out["A_m_rowsum"] = I.Asens3[ng:2*ng, rho.size:2*rho.size].sum(axis=1)
    out["A_g_colsum"] = I.Asens3[:ng, :rho.size].sum(axis=0)
    out["A_m_colsum"] = I.Asens3[ng:2*ng, rho.size:2*rho.size].sum(axis=0)
elif mode == "forward":
    import pandas as pd
    I = inv.Inversion(); I.create_cubegeometry()
    cube = pd.read_csv(job["simcube"]); sv = pd.read_csv(job["simsurvey"])
    loc = np.asarray([sv.X.values, sv.Y.values, sv.X.values*0 + cfg.zoff]).T
    Ag, _ = sm.A_sens(cfg.magneticField * 0., loc, I.Edges, 'grav')
    Am, _ = sm.A_sens(cfg.magneticField, loc, I.Edges, 'magn')
    out.update(density=cube.DENSITY.values, magsus=cube.MAGSUS.values, sensor_locations=loc,
               gravity_csv=sv.GRAVITY.values, magnetic_csv=sv.MAGNETIC.values,
               gravity_ref=Ag @ cube.DENSITY.values, magnetic_ref=Am @ cube.MAGSUS.values, Edges=I.Edges)
elif mode == "example":
    cap = {}
    orig = inv.Inversion.cubing
    def spy(self, g, m, d, loc, d0):
        cap.update(gravfield=np.array(g), magfield=np.array(m), drillfield=np.array(d),
                   sensor_locations=np.array(loc), drilldata0=np.array(d0), gp_length_in=np.array(self.gp_length, dtype=float))
        res = orig(self, g, m, d, loc, d0)
        cap.update(cubes=np.asarray(res), logl=self.logl, gp_length_out=np.array(self.gp_length, dtype=float))
        return res
    inv.Inversion.cubing = spy
    import geobo.run_geobo            # module body = the program (run_geobo.py:380-469)
    out.update(cap)
np.savez_compressed(job["out"], **{k: np.asarray(v) for k, v in out.items()})
'''

STUBS = {
    "rasterio.py": r'''
import numpy as np, struct, builtins
float32 = "float32"
class _DS:
    def __init__(self, path): self.path = path
    def read(self, band):
        b = builtins.open(self.path, "rb").read()
        e = "<" if b[:2] == b"II" else ">"
        off = struct.unpack(e + "I", b[4:8])[0]; n = struct.unpack(e + "H", b[off:off+2])[0]
        tags = {}
        for i in range(n):
            t, ty, cnt, val = struct.unpack(e + "HHI4s", b[off+2+12*i: off+14+12*i])
            sz = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 12: 8}[ty]
            if sz * cnt <= 4: data = val[: sz * cnt]
            else:
                p = struct.unpack(e + "I", val)[0]; data = b[p: p + sz * cnt]
            if ty in (3, 4): tags[t] = list(struct.unpack(e + ("H" if ty == 3 else "I") * cnt, data))
        w, h, bits, comp = tags[256][0], tags[257][0], tags[258][0], tags.get(259, [1])[0]
        assert comp == 1, "compressed tiff"
        fmt = tags.get(339, [1])[0]; assert fmt == 3
        raw = b"".join(b[o: o + c] for o, c in zip(tags[273], tags[279]))
        return np.frombuffer(raw, dtype=np.dtype(e + "f" + str(bits // 8))).reshape(h, w).astype("f" + str(bits // 8))
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def write(self, *a): pass
def open(path, *a, **k): return _DS(path)
''',
    "pyvista.py": r'''
class UniformGrid:
    def __init__(self): self.cell_arrays = {}
    def save(self, fname): pass
''',
    "skimage/__init__.py": "",
    "skimage/measure.py": "def marching_cubes_lewiner(*a, **k): raise RuntimeError('stub')\n",
}


def run_worker(job, tmp):
    jf = os.path.join(tmp, "job.json")
    json.dump(job, open(jf, "w"))
    wf = os.path.join(tmp, "worker.py")
    open(wf, "w").write(WORKER)
    r = subprocess.run([sys.executable, wf, jf], cwd=os.path.join(REF, "geobo"), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:] + r.stderr[-3000:])


def write_yaml(settings, path):
    import yaml
    os.makedirs(settings["inpath"], exist_ok=True)
    yaml.safe_dump(settings, open(path, "w"))


def make_stubs(tmp):
    d = os.path.join(tmp, "stubs")
    for rel, src in STUBS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").write(textwrap.dedent(src))
    return d


def read_vtk_cube(path):
    """legacy-binary VTK written by cubeshow.py:184-189: big-endian f8 CELL_DATA, F-order."""
    b = open(path, "rb").read()
    dims = [int(v) for v in b[b.index(b"DIMENSIONS"):].split(b"\n")[0].split()[1:4]]
    n = (dims[0] - 1) * (dims[1] - 1) * (dims[2] - 1)
    k = b.index(b"LOOKUP_TABLE default\n") + len(b"LOOKUP_TABLE default\n")
    v = np.frombuffer(b[k:k + 8 * n], dtype=">f8").astype(np.float64)
    return v.reshape([d - 1 for d in dims], order="F")


def main(which):
    tmp = tempfile.mkdtemp(prefix="geobo_golden_")
    stubdir = make_stubs(tmp)

    def job(mode, settings, out, **kw):
        y = os.path.join(tmp, "settings.yaml")
        write_yaml(settings, y)
        run_worker(dict(mode=mode, settings_yaml=y, stubdir=stubdir, out=os.path.join(HERE, out), **kw), tmp)
        print("wrote", out, flush=True)

    if "F0" in which:
        job("kat", BASE_SETTINGS, "kat_kernels.npz")
    if "F1" in which:
        for kern, gl in (("exp", None), ("sparse", None), ("matern32", [200.0, 210.0, 220.0])):
            s = dict(BASE_SETTINGS, kernelfunc=kern)
            job("cubing", s, "tiny_%s.npz" % kern, md=5, gp_length=gl, save_A=True, save_AkA=True)
        s = dict(BASE_SETTINGS, kernelfunc="exp")
        job("cubing", s, "tiny_exp_nodrill.npz", md=0, gp_length=None)
    if "F2" in which:
        cube = dict(BASE_SETTINGS, xmax=1600, ymax=1600, zLcube=1600.0, xNcube=16, yNcube=16, zNcube=16)
        job("cubing", dict(cube, kernelfunc="exp"), "cube16_exp.npz", md=0, gp_length=None)
        job("cubing", dict(cube, kernelfunc="matern32"), "cube16_matern32.npz", md=50, gp_length=[200.0, 202.0, 204.0])
        job("cubing", dict(cube, kernelfunc="sparse"), "cube16_sparse.npz", md=8, gp_length=None)
    if "F3" in which:
        import yaml
        for i, res in ((1, "cylinders"), (2, "sample")):
            s = yaml.safe_load(open(os.path.join(REF, "examples/settings_example%d.yaml" % i)))
            s.update(inpath=os.path.join(REF, "examples/testdata", "synthetic" if i == 1 else "sample") + "/",
                     outpath="/tmp/geobo_golden/out%d/" % i, gen_simulation=False, plot3d=False, plot_vertical=False,
                     bayesopt_vertical=False, bayesopt_nonvertical=False)
            name = "example%d.npz" % i
            job("example", s, name)
            d = dict(np.load(os.path.join(HERE, name)))
            names = ["cube_density", "cube_magsus", "cube_drill", "cube_density_variance", "cube_magsus_variance",
                     "cube_drill_variance"]
            d["vtk_cubes"] = np.asarray([read_vtk_cube(os.path.join(REF, "examples/results", res, n + ".vtk"))
                                         for n in names])
            d["settings_json"] = np.asarray(json.dumps({k: v for k, v in s.items() if k not in ("inpath", "outpath")}))
            np.savez_compressed(os.path.join(HERE, name), **d)
            for c, v in zip(d["cubes"], d["vtk_cubes"]):
                print("   rerun-vs-committed-VTK normwise:", np.abs(c - v).max() / np.abs(v).max())
    if "F4" in which:
        import yaml
        s = yaml.safe_load(open(os.path.join(REF, "examples/settings_example1.yaml")))
        s.update(inpath="/tmp/geobo_golden/in/", outpath="/tmp/geobo_golden/out/", gen_simulation=False)
        job("forward", s, "forward_kat.npz",
            simcube=os.path.join(REF, "examples/testdata/synthetic/simcube_cylinders.csv"),
            simsurvey=os.path.join(REF, "examples/testdata/synthetic/simsurveydata_cylinders.csv"))
        d = np.load(os.path.join(HERE, "forward_kat.npz"))
        print("   forward KAT rel err:", np.abs(d["gravity_ref"] - d["gravity_csv"]).max() / np.abs(d["gravity_csv"]).max(),
              np.abs(d["magnetic_ref"] - d["magnetic_csv"]).max() / np.abs(d["magnetic_csv"]).max())


    if "F5" in which:
        import yaml
        s = yaml.safe_load(open(os.path.join(REF, "examples/settings_example1.yaml")))
        s.update(inpath="/tmp/geobo_golden/in/", outpath="/tmp/geobo_golden/out/", gen_simulation=False, xNcube=16, yNcube=16,
                 zNcube=16, kernelfunc="exp", gp_coeff=[0.0, 0.0, 0.0], plot3d=False, plot_vertical=False,
                 bayesopt_vertical=False, bayesopt_nonvertical=False)
        job("cubing", s, "config1_exp16.npz", md=0, gp_length=None, chi_factor=0.2)
        d = dict(np.load(os.path.join(HERE, "config1_exp16.npz")))
        d["settings_json"] = np.asarray(json.dumps({k: v for k, v in s.items() if k not in ("inpath", "outpath")}))
        np.savez_compressed(os.path.join(HERE, "config1_exp16.npz"), **d)
    if "F6" in which:
        hard = dict(BASE_SETTINGS, gp_lengthscale=8, gp_err=[0.01, 0.01, 0.01])
        job("cubing", dict(hard, kernelfunc="exp"), "illcond_tiny_exp.npz", md=5, gp_length=None, gp_amp=2.0, save_A=True,
            save_AkA=True, save_Ldiag=True)
        job("cubing", dict(hard, kernelfunc="matern32", gp_lengthscale=10), "illcond_tiny_matern32.npz", md=5,
            gp_length=[1000.0, 1010.0, 1020.0], gp_amp=2.0, save_A=True, save_AkA=True, save_Ldiag=True)
        cube = dict(hard, xmax=1600, ymax=1600, zLcube=1600.0, xNcube=16, yNcube=16, zNcube=16)
        job("cubing", dict(cube, kernelfunc="matern32", gp_lengthscale=10), "illcond_cube16_matern32.npz", md=50,
            gp_length=[1000.0, 1010.0, 1020.0], gp_amp=2.0, save_Ldiag=True)
        for n in ("illcond_tiny_exp", "illcond_tiny_matern32", "illcond_cube16_matern32"):
            print("   cond(AkA) %s = %.3e" % (n, float(np.load(os.path.join(HERE, n + ".npz"))["cond_AkA"])))
    if "F7" in which:
        job("cubing", dict(BASE_SETTINGS, kernelfunc="exp"), "optimize_tiny_exp.npz", md=5, gp_length=None, optimize=True)
        d = np.load(os.path.join(HERE, "optimize_tiny_exp.npz"))
        print("   optimum", d["opt_x"], "objective", float(d["opt_fun"]))


if __name__ == "__main__":
    main(sys.argv[1:] or ["F0", "F1", "F2", "F3", "F4", "F5", "F6", "F7"])
