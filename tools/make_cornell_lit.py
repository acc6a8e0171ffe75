#!/usr/bin/env python
"""Writes assets/cornell_lit/cornell_lit.gltf: the Cornell box of assets/cornell (same .bin, same
nodes and meshes) with materials that exercise the lit integrator: the ceiling
quad emits (emissiveFactor x KHR_materials_emissive_strength), the short box is clear glass
(KHR_materials_transmission + KHR_materials_ior), the tall box is a rough metal.  The reference's
own cornell.gltf has no light source at all (its "light" quad is a 0.78 diffuse)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = json.load(open(os.path.join(ROOT, "assets", "cornell", "cornell.gltf")))
src["buffers"][0]["uri"] = "../cornell/cornell.bin"
for m in src["materials"]:
    pbr = m["pbrMetallicRoughness"]
    if m["name"] == "light":
        m["emissiveFactor"] = [1.0, 0.85, 0.6]
        m["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": 17.0}}
        pbr["baseColorFactor"] = [0.0, 0.0, 0.0, 1.0]
    elif m["name"] == "shortBox":
        pbr["baseColorFactor"] = [0.95, 0.98, 1.0, 1.0]
        pbr["roughnessFactor"] = 0.0
        m["extensions"] = {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.5}}
    elif m["name"] == "tallBox":
        pbr["baseColorFactor"] = [0.95, 0.64, 0.54, 1.0]
        pbr["metallicFactor"] = 1.0
        pbr["roughnessFactor"] = 0.35
src["extensionsUsed"] = ["KHR_materials_emissive_strength", "KHR_materials_transmission", "KHR_materials_ior"]
src.setdefault("asset", {})["extras"] = {"note": "derived from assets/cornell/cornell.gltf by tools/make_cornell_lit.py; see assets/cornell/license.txt"}
out = os.path.join(ROOT, "assets", "cornell_lit", "cornell_lit.gltf")
json.dump(src, open(out, "w"), indent=1)
print(out)
