"""Generates tests/golden/ref_geom.npz: inputs and the outputs of the REFERENCE'S OWN HEADERS (oracle/_ref/libref_geom.so, built
from /root/reference/include by oracle/Makefile `_ref` — build container only) for 1 024 primitive cases and two small scenes.
Data only: the fixture holds numbers, no source.  Run:  python tests/golden/make_ref_geom_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_geom_cases as rg  # noqa: E402

ref = rg.Ref()
rng = np.random.default_rng(77)
c = rg.primitive_cases(rng, 1024)
out = {f"in_{k}": v for k, v in c.items()}
out.update({f"out_{k}": v for k, v in ref.primitives(c).items()})
for tag, (ns, npl) in (("s0", (150, 0)), ("s1", (60, 40))):
    sph, pl, types, o, d = rg.scene_cases(rng, ns, npl, 512)
    out.update({f"{tag}_sph": sph, f"{tag}_pl": pl, f"{tag}_types": types, f"{tag}_o": o, f"{tag}_d": d})
    out.update({f"{tag}_{k}": v for k, v in ref.scene(sph, pl, types, o, d).items()})
np.savez_compressed(os.path.join(HERE, "ref_geom.npz"), **out)
print("wrote", os.path.join(HERE, "ref_geom.npz"), os.path.getsize(os.path.join(HERE, "ref_geom.npz")), "bytes")
