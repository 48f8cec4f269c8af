"""Build libzonos_hip variants with other compile-time kernel parameters into build/variants/ (git-ignored, travels with gpurun):
    python tools/build_variants.py name1=-DZN_SK_NBUF=4 name2="-DZN_SK_HELP=0 -DZN_SK_PARK=4" ...
A tool selects one with ZONOS_HIP_LIB_VARIANT=name1 (zonos_amd/_lib.py; development only).  ZN_VARIANT_SRC=zn_dac.hip applies the flags to
that source instead of zn_api.hip."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zonos_amd import build as zb  # noqa: E402

zb.build(verbose=False)
out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)


SRC = os.environ.get("ZN_VARIANT_SRC", "zn_api.hip")


def one(spec):
    name, flags = spec.split("=", 1)
    obj = os.path.join(out, f"{SRC[:-4]}_{name}.o")
    so = os.path.join(out, f"libzonos_hip_{name}.so")
    subprocess.run([zb._hipcc(), *zb.FLAGS, *flags.split(), "-c", os.path.join(zb.CSRC, SRC), "-o", obj], check=True)
    others = [os.path.join(zb.CSRC, s.replace(".hip", ".o")) for s in zb.SOURCES if s != SRC]
    subprocess.run([zb._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj, *others], check=True)
    os.remove(obj)
    return so


with ThreadPoolExecutor(max_workers=4) as ex:
    for so in ex.map(one, sys.argv[1:]):
        print(so)
