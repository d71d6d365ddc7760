"""Build A/B variants of libgeobo_hip.so (extra -D flags) into geobo_amd/lib/variants/<name>.so -- git-ignored, but shipped to the
GPU box -- and restore the default build.  Select one with GEOBO_HIP_LIB=geobo_amd/lib/variants/<name>.so.
    python tools/ab_variants.py name1=-DFLAG1,-DFLAG2 name2=-DFLAG3 ..."""
import os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobo_amd import build as B
vd = os.path.join(B.LIBDIR, "variants")
os.makedirs(vd, exist_ok=True)
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    B.build(extra_flags=tuple(f for f in flags.split(",") if f))
    shutil.copy(B.LIB, os.path.join(vd, name + ".so"))
    print("built", name, flags, flush=True)
B.build(force=False)
if open(B.STAMP).read().strip() != "default":
    os.remove(B.LIB)
    B.build()
print("default restored:", B.up_to_date())
