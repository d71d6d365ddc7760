"""Route planning: every shape / rank-count / precision decision of an inversion step as ONE pure function.

`plan_route` maps (grid extents, world size, assembly precision, operator mode, method, environment overrides) to a `Route`: which
algorithm family the engine runs (engine.py reads nothing else to decide) and which kernels carry each stage.  No device, no torch:
the whole table is tested on the CPU for every BASELINE.json configuration x world in {1, 2, 4, 8}
(tests/test_host_logic_cpu.py::test_route_table).  The behaviour-changing GEOBO_* environment switches are overrides INTO this
function; what a step then does is recorded as `engine.route` / `engine.step_route` and in the bench line.

Families (DESIGN.md sections 2 and 7):
  "rows"     the structured algorithm sharded by SENSOR ROWS over world >= 1 ranks: the three blocks of A K that AkA's lower triangle
             needs, row block by row block through the spectral product and straight on through the lattice Gram (A K is never held),
             replicated Cholesky, transposed posterior V = (L^-1 A3) K on the rank's rows of L^-1, one all-gather + one all-reduce.
             Needs a lattice survey with even stencils (decided per operator build: engine falls back to "columns").
  "single"   one rank, fp64, the fused n = 64 kernels: the same algorithm with A K materialised once (engine._posterior_zpath,
             _assemble_AkA_sym) and, with operators="auto", operator rows read as windows of the stencil table.
  "columns"  voxel-column shards of A K / V (rounds 1-2): fused reduction L^-1 (A K), AkA by the N/G-deep GEMM or the column form of
             the lattice Gram, row exchange (all-to-all) from 4 ranks.  What runs for surveys off the lattice, the dense method, padded
             shapes and small grids without fused kernels.
Reference call sites all of these replace: inversion.py:92-117 (predict3), kernels.py:158-195 (create_cov)."""
from dataclasses import asdict, dataclass

PAD_M, PAD_N = 256, 128
XZ2D_SHAPES = ((48, 64), (64, 64), (64, 32))      # fused (x, z) transform instances (hip.XZ2D_SHAPES)
XZ2D_FOLD_N = (64,)                               # radix-2 instances (hip.XZ2D_FOLD_N)
TOEPLITZ_NY = (16, 32, 48, 64, 80, 96, 112, 128)               # Toeplitz y-stage instances (hip.TOEPLITZ_NY)
SPECTRAL_AXIS_N = (80, 96, 112, 128)              # radix-4 axis passes (hip.SPECTRAL_AXIS_N); extents on the half-integer basis only
SPECTRAL_Y_NY = (32, 48, 64, 80, 96, 112, 128)    # in-kernel spectral y stage on the matrix pipe (hip.SPECTRAL_Y_NY); > 64: geobo_spectral_y3 only
ROWS_MIN_VOXELS = 1 << 18                         # batched-GEMM forms of the row algorithm pay from 64^3 voxels ...
ROWS_MIN_PLANE = 96 * 96                          # ... and (x, z) planes that fill the 128 x 128 GEMM tiles
ROWS_MIN_VOXELS_MID, ROWS_MIN_PLANE_MID = 3 << 17, 64 * 64   # ... or from 393 216 voxels with planes of 4096 modes: 80^3 3345 / 4733 ms (52 / 183 GB),
                                                  # 80x64x80 2026 / 2503; at the old threshold 96x32x96 786 / 741; below: 80x32x80 528 / 414, 112x16x112 290 / 212,
                                                  # 16x128x128 462 / 367 (2048-mode planes), 48^3 237 / 188, 32x32x128 100 / 63
COLUMN_FORM_MAX_BYTES = 160 << 30                 # a two-property A K of the column form beyond this does not fit beside its workspaces (80x128x80:
                                                  # 272 GB, out of memory; row form 8.8 s in 63 GB)
ROWS_MIN_VOXELS_FUSED = 1 << 17                   # with fused / four-plane (x, z) kernels and the Toeplitz y stage: from 2^17 voxels.  Measured
                                                  # (row form / column form, ms per step, one MI355X): 64x32x64 107 / 138, 32x128x32 379 / 482,
                                                  # 64x64x32 369 / 484, 48x64x64 341 / 424, 64x80x64 877 / 1736, 64x128x64 2365 / 6670 (50 / 235 GB);
                                                  # below: 64x32x32 101 / 80, 48x32x64 89 / 69, 32x64x32 87 / 77, 64x16x64 35 / 33


# Every behaviour switch of a step, in ONE table: option -> the environment override that turns it off ("0").  plan_route is the only
# reader of the environment; the engine hands the resolved options on (Route.opts()) to the spectral product, the lattice Gram and the
# transposed posterior, which used to read os.environ themselves (round-5 review, item 8).  All of them are A/B and fallback-coverage
# switches: the default of every option is "on", and every "off" path is a complete, tested form of the same arithmetic.
SWITCHES = {"fused_xz": "GEOBO_SPECTRAL_FUSED_XZ",   # fused (x, z) transform kernels (else two batched passes)
            "fold": "GEOBO_XZ_FOLD",                 # radix-2 / radix-4 transform kernels (else the plain products)
            "quad": "GEOBO_XZ_QUAD",                 # 32 x 32 planes four at a time through the n = 64 radix-2 kernels (else stacked pairs)
            "dense_y": "GEOBO_SPECTRAL_DENSE_Y",     # y axis applied per mode inside one kernel (else carried through the spectrum by passes)
            "y_mfma": "GEOBO_Y_MFMA",                # ... as an in-kernel spectral product on the matrix pipe (else the direct vector-pipe kernels)
            "axis_mfma": "GEOBO_AXIS_MFMA",          # x passes of the unfused (x, z) transforms as radix-4 axis kernels (else radix-2 GEMM passes)
            "y2s": "GEOBO_Y2S",                      # two-term rows with the shared cross block K_01 = K_10 (else four products)
            "z_fused": "GEOBO_Z_FUSED",              # rows of L^-1 A: one fused inverse transform per (row, z) plane (else two GEMM passes)
            "z_mul": "GEOBO_Z_MUL",                  # ... its input product formed inside the kernel (else written and read back)
            "z_lattice": "GEOBO_Z_LATTICE",          # rows of L^-1 A as lattice convolutions (else triangular GEMMs against the operator)
            "aka_lattice": "GEOBO_AKA_LATTICE"}      # AkA by the lattice Gram (else the N-deep GEMM)


def switches(env=None):
    """{option: bool} from an environment-like mapping (None: every option on)."""
    env = {} if env is None else env
    return {k: env.get(v, "1") != "0" for k, v in SWITCHES.items()}


def _pad(v, m):
    return (int(v) + m - 1) // m * m


def shard_columns(n_pad, world, rank):
    units = n_pad // PAD_N
    return units * rank // world * PAD_N, units * (rank + 1) // world * PAD_N


@dataclass(frozen=True)
class Route:
    spectral: bool          # A K through the real-DFT route (regular grid, extents % 16, unpadded voxels, slab-aligned shards)
    family: str             # "rows" | "single" | "columns": what the engine attempts (a survey off the lattice demotes to "columns")
    rows: bool              # the row form is statically possible
    single: bool            # the one-rank fused-kernel form is statically possible
    exchange: bool          # column form: row-sharded transforms + all-to-all of A K block columns
    exchange_without_rows: bool   # ... when the row form is denied at operator-build time (off-lattice survey, uneven stencil)
    operators: str          # "resident" | "streamed" | "auto" as requested, resolved to "streamed" where residency cannot fit
    kernels: tuple          # (("xz", ...), ("y", ...), ("gram", ...), ("ss", ...)): which implementation carries each stage
    note: str               # why a faster family stepped aside for this shape ("" when nothing did)
    ak_bytes: int = 0       # what a materialised A K (column form / one-rank fused form) of this rank would take
    rows_mandatory: bool = False   # ak_bytes > COLUMN_FORM_MAX_BYTES: only the row form fits; a denied row form is an error, not a fallback
    switches: tuple = ()    # ((option, bool), ...) of SWITCHES as resolved from the environment handed to plan_route

    def opt(self, name):
        return dict(self.switches).get(name, True)

    def opts(self):
        return dict(switches(None), **dict(self.switches))

    def describe(self):
        k = dict(self.kernels)
        return "%s/%s xz=%s y=%s gram=%s ss=%s%s" % ("spectral" if self.spectral else "dense", self.family, k["xz"], k["y"], k["gram"],
                                                      k["ss"], " exchange" if self.exchange and self.family == "columns" else "")

    def as_dict(self):
        d = asdict(self)
        d["kernels"] = dict(self.kernels)
        d["switches"] = dict(self.switches)
        return d


def lattice_gram_fast(nx, ny, nz):
    """Grids whose Gram x step / back-transform run on the fused kernels (lattice_gram.LatticeGram.fast)."""
    return nx == 64 and nz == 64 and ny in (48, 64)


def lattice_gram_supported(nx, ny, nz):
    return nx % 16 == 0 and ny % 16 == 0 and nz % 16 == 0 and ny >= 16


def plan_route(nx, ny, nz, world=1, rank=0, assembly="f64", operators="resident", method="auto", env=None, nprops=2):
    env = {} if env is None else env
    sw = switches(env)
    nx, ny, nz, world = int(nx), int(ny), int(nz), int(world)
    N, Ms = nx * ny * nz, nx * ny
    N_pad, Ms_pad = _pad(N, PAD_N), _pad(Ms, PAD_M)
    plane = nx * nz
    f32, streamed = assembly == "f32", operators == "streamed"
    notes = []
    # slab alignment is decided over ALL ranks' shards (round-4 advisory: decided per rank, 16^3 on 3 / 5 / 6 / 7 ranks gave some ranks
    # the spectral route and others the dense one -- same collectives, different arithmetic and bench lines per rank)
    shards = [shard_columns(N_pad, world, r) for r in range(world)]
    aligned = all(c0 % plane == 0 and c1 % plane == 0 and c1 > c0 for c0, c1 in shards)
    spectral = (method in ("auto", "spectral") and nx % 16 == 0 and ny % 16 == 0 and nz % 16 == 0 and N == N_pad and aligned)
    fused_xz = (nx, nz) in XZ2D_SHAPES and sw["fused_xz"]
    pair_xz = (nx, nz) == (32, 32) and (64, 32) in XZ2D_SHAPES and ny % 2 == 0 and sw["fused_xz"]
    fold = sw["fold"] and nx == nz and nx in XZ2D_FOLD_N
    dense_y = ny in TOEPLITZ_NY and sw["dense_y"]
    y_mfma = dense_y and ny in SPECTRAL_Y_NY and sw["y_mfma"]
    fused_ss = fused_xz and fold and dense_y          # (any Toeplitz ny: the reduction in the inverse transform takes any plane count)
    transposed = env.get("GEOBO_POSTERIOR", "zpath") == "zpath"
    unpadded = Ms == Ms_pad and N == N_pad
    gram_ok = lattice_gram_supported(nx, ny, nz) and sw["aka_lattice"]
    gram_fast = lattice_gram_fast(nx, ny, nz)
    single = world == 1 and not f32 and spectral and unpadded and transposed and fused_ss
    quad_xz = pair_xz and ny % 4 == 0 and sw["fold"] and 64 in XZ2D_FOLD_N and sw["quad"]
    rows_mode = env.get("GEOBO_ROWS", "auto")           # "0": never; "1": wherever it is possible; "auto": where it pays
    # a materialised A K of this rank: nprops property blocks in the element size of the assembly (fp32 assembly stores A K as fp32); the
    # engine repeats the check with the step's own property count where it allocates (engine._assemble_AK)
    column_ak_bytes = (2 * Ms_pad + PAD_M) * int(nprops) * N_pad * (4 if f32 else 8) // max(world, 1)
    pays = ((gram_fast and fused_ss) or ((fused_xz or quad_xz) and dense_y and N >= ROWS_MIN_VOXELS_FUSED)
            or (N >= ROWS_MIN_VOXELS and plane >= ROWS_MIN_PLANE) or (N >= ROWS_MIN_VOXELS_MID and plane >= ROWS_MIN_PLANE_MID)
            or column_ak_bytes > COLUMN_FORM_MAX_BYTES)
    xmode = env.get("GEOBO_SPECTRAL_EXCHANGE", "auto")    # "0": replicated forward transforms, column shards (also switches the row form off for N > 1)
    rows = (spectral and unpadded and Ms % world == 0 and gram_ok and transposed and sw["z_lattice"] and rows_mode != "0"
            and (pays or rows_mode == "1") and not (single and gram_fast and rows_mode != "1") and (world == 1 or xmode != "0" or rows_mode == "1"))
    ncs = {shard_columns(N_pad, world, r)[1] - shard_columns(N_pad, world, r)[0] for r in range(world)}
    xbase = spectral and world > 1 and Ms % world == 0 and len(ncs) == 1 and xmode != "0"
    exchange_without_rows = xbase and (world >= 4 or xmode == "1")
    exchange = xbase and (world >= 4 or xmode == "1" or rows)
    family = "rows" if rows else ("single" if single else "columns")
    if spectral and family == "columns" and transposed:
        if not unpadded:
            notes.append("nx*ny = %d is not a multiple of %d (padded sensor rows): fused reduction L^-1 (A K) instead of the transposed order" % (Ms, PAD_M))
        elif not pays:
            notes.append("the structured algorithm pays from %d voxels with fused (x, z) kernels and from %d voxels with (x, z) planes of %d "
                         "modes (%d with %d) on the batched-GEMM forms (here %d x %d x %d: %d voxels, %d modes, %s (x, z) kernels): N-deep Gram "
                         "and fused reduction (2-6x the work of the structured forms at larger sizes)"
                         % (ROWS_MIN_VOXELS_FUSED, ROWS_MIN_VOXELS, ROWS_MIN_PLANE, ROWS_MIN_VOXELS_MID, ROWS_MIN_PLANE_MID, nx, ny, nz, N, plane,
                            "fused" if (fused_xz or quad_xz) else "no fused"))
        elif world > 1 and Ms % world:
            notes.append("%d sensor rows do not divide over %d ranks: column shards" % (Ms, world))
    if not spectral and method == "auto" and (nx % 16 or ny % 16 or nz % 16):
        notes.append("grid extents %d x %d x %d are not multiples of 16: dense route (2 Ms N^2 flop per block pair)" % (nx, ny, nz))
    # operator residency: one operator is Ms_pad x N_pad doubles; in the row form a rank only ever needs its rows + two boundary slabs
    ops = operators
    rows_r = Ms // world if Ms % world == 0 else Ms
    if family == "rows" and not streamed and rows_r * N_pad * 8 > (40 << 30):
        ops = "streamed"
        notes.append("operator rows of a rank (%.0f GB) are generated per batch instead of being resident" % (rows_r * N_pad * 8 / 1e9))
    kernels = (("xz", "fold" if (fused_xz and fold) else "fused" if fused_xz else "quad" if quad_xz else "pair" if pair_xz else
                "gemm+axis4" if (nx in SPECTRAL_AXIS_N and sw["axis_mfma"]) else "gemm"),
               ("y", ("mfma" if y_mfma else "toeplitz") if dense_y else "spectrum"),
               ("gram", ("fused" if gram_fast else "gemm") if (family in ("rows", "single") and gram_ok) else "per-step"),
               ("ss", ("fused" if fused_ss else "stored") if family in ("rows", "single") else "reduction"))
    if column_ak_bytes > COLUMN_FORM_MAX_BYTES and family != "rows":
        notes.append("a materialised A K of this grid takes %.0f GB per rank (limit %.0f): only the row form fits, and it is %s"
                     % (column_ak_bytes / 1e9, COLUMN_FORM_MAX_BYTES / 1e9, "switched off (GEOBO_ROWS=0)" if rows_mode == "0" else "not available here"))
    return Route(spectral=spectral, family=family, rows=rows, single=single, exchange=exchange, exchange_without_rows=exchange_without_rows,
                 operators=ops, kernels=kernels, note="; ".join(notes), ak_bytes=int(column_ak_bytes),
                 rows_mandatory=bool(column_ak_bytes > COLUMN_FORM_MAX_BYTES), switches=tuple(sorted(sw.items())))
