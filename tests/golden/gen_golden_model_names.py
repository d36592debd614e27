#!/usr/bin/env python3
"""Golden for the model-merge order: the body / geom / joint name order and the object geoms' contact attributes of the ONE merged
model the reference keeps in its tree, /root/reference/dataset_model_temp.xml (the Banana model MujocoXML.merge wrote: SURVEY.md
§2 row 20), in MuJoCo's compile order (bodies depth-first in document order; a body's geoms in document order when the body is
visited).  Data only: names and attribute values, no source text.  Run in the build container:
    python3 tests/golden/gen_golden_model_names.py [/root/reference]"""
import json, os, sys
import xml.etree.ElementTree as ET

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
root = ET.parse(os.path.join(ref, "dataset_model_temp.xml")).getroot()
bodies, geoms, joints, obj_geoms = ["world"], [], [], {}


def visit(b, name):
    for ch in b:
        if ch.tag == "geom":
            geoms.append(ch.get("name"))
            if name == "banana":
                obj_geoms[ch.get("name")] = {k: ch.get(k) for k in ("condim", "density", "friction", "solref", "solimp", "contype", "conaffinity", "mesh")
                                             if ch.get(k) is not None}
        elif ch.tag == "joint":
            joints.append({"name": ch.get("name"), "type": ch.get("type", "hinge"), "body": name})
    for ch in b:
        if ch.tag == "body":
            bodies.append(ch.get("name"))
            visit(ch, ch.get("name"))


visit(root.find("worldbody"), "world")
size = root.find("size")
out = {"source": "dataset_model_temp.xml", "bodies": bodies, "geoms": geoms, "joints": joints, "banana_geoms": obj_geoms,
       "size": dict(size.attrib) if size is not None else {}}
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_model_temp_names.json")
json.dump(out, open(p, "w"), indent=1)
print(p, len(bodies), "bodies", len(geoms), "geoms", len(joints), "joints", out["size"])
