"""geobo_amd -- MI355X-native (gfx950) implementation of GeoBO's GP joint-inversion hot path.

Public surface = the reference's three hot-path modules, same names and signatures:
    geobo_amd.kernels      (geobo/kernels.py)      covariance library, create_cov
    geobo_amd.sensormodel  (geobo/sensormodel.py)  gravity / magnetic forward operators, drill selection
    geobo_amd.inversion    (geobo/inversion.py)    class Inversion: cubing(), predict3(), calc_logl(), optimize_gp()
    geobo_amd.config_loader                        the YAML settings the reference keeps in module globals

Underneath: hand-written HIP kernels behind a C ABI (include/geobo_hip.h, geobo_amd/csrc), loaded with ctypes.
Importing the package never needs a GPU; calling a compute function without the HIP extension or a device raises.
"""
__version__ = "0.1.0"

from . import config_loader  # noqa: F401
