#!/usr/bin/env python3
"""Converts the in-scope reference MJCF files into the neutral .cmodel text form.

usage: make_models.py <reference model dir> <output dir> [<test fixture dir>]

The .cmodel files are the compiled *data* of model/cassie.xml, cassie_hfield.xml and
cassie_tray_box.xml as produced by this repo's own MJCF loader (mjcf_loader.cpp); they
are committed so that boxes without /root/reference (the GPU box) can load the models.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cassie-mujoco-sim_amd"))
from cassie_amd import Model  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for name in ("cassie", "cassie_hfield", "cassie_tray_box"):
    m = Model(os.path.join(src, name + ".xml"))
    out = os.path.join(dst, name + ".cmodel")
    m.save(out)
    print("wrote", out, "nq", m.pod.nq, "nv", m.pod.nv, "nbody", m.pod.nbody)

# test fixture: cassie.xml with a second hinge on the left plantar rod's body -- a body with two rotational joints, i.e. a
# model that is NOT kin_simple (cm_model.h) and therefore runs the run-time-topology kernel with the general joint loop
if len(sys.argv) > 3:
    import tempfile
    xml = open(os.path.join(src, "cassie.xml")).read()
    i = xml.index("<joint name='left-plantar-rod'")
    j = xml.index("/>", i) + 2
    xml = xml[:j] + "\n<joint name='left-extra' type='hinge' axis='0 1 0' pos='0.01 0 0.02' limited='false'/>" + xml[j:]
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "cassie_two_hinges.xml"), "w").write(xml)
        os.symlink(os.path.join(os.path.abspath(src), "cassie-stl-meshes"), os.path.join(tmp, "cassie-stl-meshes"))
        m = Model(os.path.join(tmp, "cassie_two_hinges.xml"))
        out = os.path.join(sys.argv[3], "cassie_two_hinges.cmodel")
        m.save(out)
        print("wrote", out, "nq", m.pod.nq, "nv", m.pod.nv, "kin_simple", m.pod.kin_simple)
