#!/usr/bin/env python3
"""Converts the in-scope reference MJCF files into the neutral .cmodel text form.

usage: make_models.py <reference model dir> <output dir>

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
