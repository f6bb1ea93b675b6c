"""Height-field contact path (BASELINE config 4, cassie_hfield.xml): the terrain of reference
example/test_hfield.py:39-41 (random heights, central 10 x 10 samples zeroed), oracle sanity and parity of
the emulated kernel."""
import numpy as np
import pytest

import oracle_py
from cassie_amd import Model
from emu_py import EmuBatch
from oracle_py import Oracle


def terrain(flat=False):
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0
    if flat:
        h[:] = 0
    return h


@pytest.fixture(scope="module")
def hf(built):
    return Model("cassie_hfield")


def test_flat_heightfield_behaves_like_the_floor_plane(hf, cassie):
    """All-zero samples: the terrain is the plane z = -0.1 (geom pos), so the robot dropped from the init pose
    must behave exactly like cassie.xml dropped on a floor moved to z = -0.1."""
    oracle_py.set_hfield(terrain(flat=True))
    try:
        o = Oracle(hf.pod, hf.qpos_init())
        from cassie_amd._lib import CmModel
        pl = CmModel.from_buffer_copy(cassie.pod)
        pl.geom_pos[0][2] = -0.1                                 # floor plane is collision geom 0 of cassie.xml
        p = Oracle(pl, cassie.qpos_init())
        for _ in range(400):
            o.step()
            p.step()
        assert o.d.ncon > 0
        assert np.max(np.abs(o.qpos - p.qpos)) < 1e-9
    finally:
        oracle_py.set_hfield(None)


def test_emulated_kernel_matches_oracle_on_rough_terrain(hf):
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        pod = hf.pod
        q = hf.qpos_init()
        q[0] = 0.35                                              # start over the rough part, straddling the flat patch edge
        o = Oracle(pod, q)
        emu = EmuBatch(pod, 1)
        emu.qpos[:] = q
        emu.hfield = h.ravel().copy()
        seen = 0
        for s in range(500):
            emu.step()
            o.step()
            assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
            seen = max(seen, o.d.ncon)
        assert seen >= 2
        assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8
        nz = [o.d.contact[i].frame[2] for i in range(o.d.ncon)]
        assert all(0 < z <= 1 for z in nz)
    finally:
        oracle_py.set_hfield(None)


def test_no_terrain_data_means_no_contact(hf):
    oracle_py.set_hfield(None)
    o = Oracle(hf.pod, hf.qpos_init())
    o.step(300)
    assert o.d.ncon == 0 and o.qpos[2] < 0.9                     # falls freely without samples


def _probe(hf, ps, r):
    import ctypes
    L = oracle_py.lib()
    L.co_test_hfield_sphere.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    out = np.zeros(7)
    p = np.ascontiguousarray(ps, dtype=np.float64)
    n = L.co_test_hfield_sphere(ctypes.byref(hf.pod), p.ctypes.data, r, 0.0, out.ctypes.data)
    return (n, out[0], out[1:4], out[4:7])


def test_step_terrain_rim_and_wall_are_felt(hf):
    """Advisor finding (round 1): a sample sphere used to see only the triangle under its centre, so a taller
    neighbouring cell was invisible.  A mesa (elevation 0.2 for x >= 0.15, else 0; the height field geom sits at z = -0.1, so
    the low ground is world z = -0.1 and the top z = +0.1) against spheres: beside the rim, over the low ground, on top, and
    sunk into the low ground."""
    nc = 200
    xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
    h = np.zeros((200, 200), dtype=np.float32)
    h[:, xs >= 0.15] = 1.0
    jr = int(np.argmax(xs >= 0.15))                  # first raised column; the slope runs from xs[jr - 1] to xs[jr]
    oracle_py.set_hfield(h)
    try:
        # over the low ground, far from the mesa: plain vertical contact
        n, dist, pos, nrm = _probe(hf, [-0.5, 0.0, -0.1 + 0.03], 0.05)
        assert n == 1 and abs(dist - (-0.02)) < 1e-12 and np.allclose(nrm, [0, 0, 1])
        # above the low ground but within reach of the slope's rim only: the old definition found nothing here
        c = np.array([xs[jr] - 0.09, 0.0, 0.1 + 0.02])  # level with the mesa top, 0.09 m before it: low ground is 0.22 m below
        n, dist, pos, nrm = _probe(hf, c, 0.1)
        assert n == 1
        rim = np.array([xs[jr], 0.0, 0.1])
        rim_dist = np.linalg.norm(c - rim) - 0.1                         # the rim of the raised cells; the steep face just below
        assert rim_dist - 1e-3 < dist <= rim_dist + 1e-12               # it (5 cm run, 20 cm rise) is a hair closer still
        assert nrm[0] < -0.9 and nrm[2] > 0                              # pushed back, away from the mesa
        # on top of the mesa
        n, dist, pos, nrm = _probe(hf, [1.0, 0.3, 0.1 + 0.04], 0.05)
        assert n == 1 and abs(dist + 0.01) < 1e-12 and np.allclose(nrm, [0, 0, 1])
        # centre below the surface of the low ground: penetration along the surface normal
        n, dist, pos, nrm = _probe(hf, [-0.5, 0.2, -0.1 - 0.01], 0.03)
        assert n == 1 and abs(dist - (-0.04)) < 1e-12 and np.allclose(nrm, [0, 0, 1])
        # high above everything: culled
        assert _probe(hf, [0.0, 0.0, 0.5], 0.05)[0] == 0
    finally:
        oracle_py.set_hfield(None)


def test_emulated_kernel_matches_oracle_at_a_terrain_step(hf):
    """Robot dropped with one foot over a 6 cm ledge: kernel (emulated) and oracle agree contact for contact."""
    nc = 200
    xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
    ys = xs
    h = np.zeros((200, 200), dtype=np.float32)
    h[ys >= 0.05, :] = 0.3                              # left foot (y = +0.135) lands on a ledge 0.06 m high
    oracle_py.set_hfield(h)
    try:
        pod = hf.pod
        q = hf.qpos_init()
        o = Oracle(pod, q)
        emu = EmuBatch(pod, 1)
        emu.qpos[:] = q
        emu.hfield = h.ravel().copy()
        for s in range(400):
            emu.step()
            o.step()
            assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
        assert o.d.ncon >= 2
        assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8
        zs = sorted(o.d.contact[i].pos[2] for i in range(o.d.ncon))
        assert zs[-1] - zs[0] > 0.04                    # contacts on both levels
    finally:
        oracle_py.set_hfield(None)


def test_capsule_feels_a_bump_under_its_middle(hf):
    """A horizontal capsule (radius 2 cm, 16 cm long, like Cassie's foot) over flat ground with one raised sample under its
    middle: two end spheres alone would report nothing; the interior samples find the bump.  Without the bump the capsule is
    treated exactly like against a plane: its two end spheres."""
    import ctypes
    L = oracle_py.lib()
    L.co_test_hfield_capsule.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    nc = 200
    xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
    j = int(np.argmin(np.abs(xs - 1.0)))
    i = int(np.argmin(np.abs(xs - 0.5)))
    h = np.zeros((200, 200), dtype=np.float32)
    h[i, j] = 0.25                                   # a 5 cm spike at (xs[j], xs[i]); ground at world z = -0.1
    mc = np.array([0.0, 0, 1, 0, 1, 0, -1, 0, 0])    # capsule axis (third column) along world x
    out = np.zeros(14)

    def probe(z):
        pc = np.array([xs[j], xs[i], z])
        n = L.co_test_hfield_capsule(ctypes.byref(hf.pod), pc.ctypes.data, mc.ctypes.data, 0.02, 0.08, 0.0, out.ctypes.data)
        return n, out.reshape(2, 7).copy()
    oracle_py.set_hfield(h)
    try:
        n, c = probe(-0.1 + 0.05 + 0.015)             # 1.5 cm above the spike's tip minus radius: touches the spike only
        assert n == 1 and abs(c[0, 0] - (-0.005)) < 1e-9
        assert abs(c[0, 1] - xs[j]) < 1e-6 and c[0, 6] > 0.99          # at the middle, pushing up
        n, c = probe(-0.1 + 0.015)                    # pressed down to the ground: the spike (deepest) and one end
        assert n == 2 and min(c[0, 0], c[1, 0]) < -0.04
        xs_contact = sorted([c[0, 1], c[1, 1]])
        assert abs(xs_contact[0] - xs[j]) < 0.03 or abs(xs_contact[1] - xs[j]) < 0.03
        oracle_py.set_hfield(np.zeros((200, 200), dtype=np.float32))
        n, c = probe(-0.1 + 0.015)
        assert n == 2 and np.allclose(c[:, 0], -0.005) and np.allclose(sorted(c[:, 1]), [xs[j] - 0.08, xs[j] + 0.08])   # the two ends
    finally:
        oracle_py.set_hfield(None)


def _closest_on_triangle_np(p, a, b, c):
    """Independent formulation: plane projection if it falls inside the triangle, else the closest point of the three edges."""
    n = np.cross(b - a, c - a)
    n /= np.linalg.norm(n)
    q = p - np.dot(p - a, n) * n
    inside = all(np.dot(np.cross(v1 - v0, q - v0), n) >= 0 for v0, v1 in ((a, b), (b, c), (c, a)))
    if inside:
        return q
    best, bd = None, 1e300
    for v0, v1 in ((a, b), (b, c), (c, a)):
        t = np.clip(np.dot(p - v0, v1 - v0) / np.dot(v1 - v0, v1 - v0), 0, 1)
        x = v0 + t * (v1 - v0)
        d = np.linalg.norm(p - x)
        if d < bd:
            best, bd = x, d
    return best


def test_sphere_distance_equals_brute_force_distance_to_the_triangulated_surface(hf):
    """On the random terrain of the reference's test_hfield.py: for sample spheres above the surface the reported distance
    is the brute-force minimum over ALL triangles of a generous neighbourhood (independent point-triangle routine), the
    normal points from the closest surface point to the centre, the contact point lies midway."""
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        nc = 200
        xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
        dx = xs[1] - xs[0]
        Z = 0.2 * h.astype(np.float64) - 0.1              # world heights: elevation * size[2] + geom z
        rng = np.random.default_rng(31)
        hits = 0
        for trial in range(150):
            r = rng.choice([0.02, 0.04, 0.08, 0.15])
            cx, cy = rng.uniform(-2, 2, 2)
            j, i = int((cx + 5) / dx), int((cy + 5) / dx)
            ztop = Z[i - 4:i + 6, j - 4:j + 6].max()
            c = np.array([cx, cy, ztop + rng.uniform(0.0, 1.2) * r])      # centre above every nearby vertex: the "above" branch
            best, bq = 1e300, None
            for ii in range(i - 5, i + 6):
                for jj in range(j - 5, j + 6):
                    v00, v10 = np.array([xs[jj], xs[ii], Z[ii, jj]]), np.array([xs[jj + 1], xs[ii], Z[ii, jj + 1]])
                    v01, v11 = np.array([xs[jj], xs[ii + 1], Z[ii + 1, jj]]), np.array([xs[jj + 1], xs[ii + 1], Z[ii + 1, jj + 1]])
                    for tri in ((v00, v10, v01), (v11, v01, v10)):
                        q = _closest_on_triangle_np(c, *tri)
                        d = np.linalg.norm(c - q)
                        if d < best:
                            best, bq = d, q
            n, dist, pos, nrm = _probe(hf, c, r)
            if best - r > 0:
                assert n == 0
                continue
            hits += 1
            assert n == 1 and abs(dist - (best - r)) < 1e-9, (trial, dist, best - r)
            assert np.allclose(nrm, (c - bq) / best, atol=1e-7)
            assert np.allclose(pos, c - nrm * (r + 0.5 * dist), atol=1e-12)
        assert hits > 40
    finally:
        oracle_py.set_hfield(None)


def _dense_model():
    from cassie_amd import phys as P
    m = Model("cassie_hfield")
    m.set_flag(P.FLAG_HFDENSE, True)
    assert m.pod.flags & P.FLAG_HFDENSE
    return m


def test_dense_sampling_flag_finds_a_spike_between_the_default_samples(hf):
    """CM_FLAG_HFDENSE (extension, off by default): a shin-like capsule (radius 4 cm, 43 cm long) lying 7 cm above flat ground
    over a 10 cm spike placed midway between two of its default sample spheres (which sit 8.7 cm apart on the 5 cm grid):
    the default sampling passes over the spike (its nearest sample spheres clear the spike's flanks), the dense sampling
    (4.8 cm apart) has a sample within 2.4 cm of the tip and reports the contact."""
    import ctypes
    L = oracle_py.lib()
    L.co_test_hfield_capsule.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    nc = 200
    xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
    j, i = int(np.argmin(np.abs(xs - 1.0))), int(np.argmin(np.abs(xs - 0.5)))
    h = np.zeros((200, 200), dtype=np.float32)
    h[i, j] = 0.5                                    # 10 cm spike (size z = 0.2) at (xs[j], xs[i]); ground at world z = -0.1
    mc = np.array([0.0, 0, 1, 0, 1, 0, -1, 0, 0])    # capsule axis along world x
    half, r = 0.215, 0.04
    spacing = 2 * half / 5                           # default: 4 interior samples -> 5 intervals of 8.6 cm
    pc = np.array([xs[j] - half + 1.5 * spacing, xs[i], -0.1 + 0.07 + r])      # spike midway between interior samples 1 and 2
    out = np.zeros(14)
    dense = _dense_model()
    oracle_py.set_hfield(h)
    try:
        n_default = L.co_test_hfield_capsule(ctypes.byref(hf.pod), pc.ctypes.data, mc.ctypes.data, r, half, 0.0, out.ctypes.data)
        n_dense = L.co_test_hfield_capsule(ctypes.byref(dense.pod), pc.ctypes.data, mc.ctypes.data, r, half, 0.0, out.ctypes.data)
        assert n_default == 0 and n_dense == 1
        assert -0.04 < out[0] < 0 and abs(out[1] - xs[j]) < 0.03 and out[6] > 0.5       # at the spike, pushing up
    finally:
        oracle_py.set_hfield(None)


def test_emulated_kernel_matches_oracle_with_dense_sampling(built):
    """The dense-sampling pre-pass (ten lanes per pair, six pairs per wave pass, two passes for Cassie's nine height-field
    pairs) against the oracle with the same flag, a robot falling over on the rough part of the terrain (shins and tarsi reach
    the ground: the long capsules are what the flag changes)."""
    m = _dense_model()
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        pod = m.pod
        q = m.qpos_init()
        q[0], q[1], q[2] = 0.6, 0.9, 0.75                        # over the rough part, low, and tipped over:
        q[3:7] = [0.924, 0.0, 0.383, 0.0]                        # pitched 45 degrees
        o = Oracle(pod, q)
        emu = EmuBatch(pod, 1)
        emu.qpos[:] = q
        emu.hfield = h.ravel().copy()
        most = 0
        for s in range(400):
            emu.step()
            o.step()
            assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
            most = max(most, o.d.ncon)
        assert most >= 4                                         # several capsules on the ground
        assert np.max(np.abs(emu.qpos[0] - o.qpos) / np.maximum(1, np.abs(o.qpos))) < 1e-7
    finally:
        oracle_py.set_hfield(None)


def _flag_model(*flags):
    m = Model("cassie_hfield")
    for f in flags:
        m.set_flag(f, True)
    return m


def test_multi_contact_flag_reports_the_deepest_samples(hf):
    """CM_FLAG_HFMULTI (extension, off by default): a foot-like capsule pressed flat into flat ground reports four contacts
    instead of its two ends -- all at the same depth, so in sample order: +h end, -h end, then the interior samples from the
    +h side; over a spike the spike's sample comes first (deepest first)."""
    import ctypes
    from cassie_amd import phys as P
    L = oracle_py.lib()
    L.co_test_hfield_capsule.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    multi = _flag_model(P.FLAG_HFMULTI)
    nc = 200
    xs = -5 + 10.0 * np.arange(nc) / (nc - 1)
    j, i = int(np.argmin(np.abs(xs - 1.0))), int(np.argmin(np.abs(xs - 0.5)))
    mc = np.array([0.0, 0, 1, 0, 1, 0, -1, 0, 0])    # capsule axis along world x
    out = np.zeros(28)
    pc = np.array([xs[j], xs[i], -0.1 + 0.015])      # radius 2 cm, centre 1.5 cm above the ground: 5 mm deep
    oracle_py.set_hfield(np.zeros((200, 200), dtype=np.float32))
    try:
        n = L.co_test_hfield_capsule(ctypes.byref(multi.pod), pc.ctypes.data, mc.ctypes.data, 0.02, 0.08, 0.0, out.ctypes.data)
        c = out.reshape(4, 7)
        assert n == 4 and np.allclose(c[:, 0], -0.005)
        assert np.allclose(c[:, 1], [xs[j] + 0.08, xs[j] - 0.08, xs[j] + 0.04, xs[j]])      # 3 interior samples: +4 cm, 0, -4 cm; the first two make the cut
        assert L.co_test_hfield_capsule(ctypes.byref(hf.pod), pc.ctypes.data, mc.ctypes.data, 0.02, 0.08, 0.0, out.ctypes.data) == 2
        h = np.zeros((200, 200), dtype=np.float32)
        h[i, j] = 0.25                               # 5 cm spike under the middle
        oracle_py.set_hfield(h)
        n = L.co_test_hfield_capsule(ctypes.byref(multi.pod), pc.ctypes.data, mc.ctypes.data, 0.02, 0.08, 0.0, out.ctypes.data)
        c = out.reshape(4, 7)
        assert n == 4 and c[0, 0] < -0.04 and abs(c[0, 1] - xs[j]) < 0.03 and np.all(np.diff(c[:, 0]) >= 0)    # deepest first
    finally:
        oracle_py.set_hfield(None)


@pytest.mark.parametrize("dense", [False, True])
def test_emulated_kernel_matches_oracle_with_multi_contact(built, dense):
    """The multi-contact rule (rank of a sample among its pair's samples = its slot of the pair's record; third and fourth
    contact straight from the LDS table to the contact list) against the oracle: a robot dropped on the flat patch (standing:
    four contacts per foot) and one tipped over on the rough part."""
    from cassie_amd import phys as P
    m = _flag_model(*([P.FLAG_HFMULTI] + ([P.FLAG_HFDENSE] if dense else [])))
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        pod = m.pod
        q0 = m.qpos_init()
        q1 = m.qpos_init()
        q1[0], q1[1], q1[2] = 0.6, 0.9, 0.75
        q1[3:7] = [0.924, 0.0, 0.383, 0.0]
        orcs = [Oracle(pod, q0), Oracle(pod, q1)]
        emu = EmuBatch(pod, 2)
        emu.qpos[0], emu.qpos[1] = q0, q1
        emu.hfield = h.ravel().copy()
        most = [0, 0]
        for s in range(350):
            emu.step()
            for e, o in enumerate(orcs):
                o.step()
                assert (emu.info[e, 0], emu.info[e, 1]) == (o.d.ncon, o.d.nefc), (s, e)
                most[e] = max(most[e], o.d.ncon)
        assert most[0] >= 6 and most[1] >= 5         # more than the two-per-capsule rule gives a standing / lying robot
        for e, o in enumerate(orcs):
            assert np.max(np.abs(emu.qpos[e] - o.qpos) / np.maximum(1, np.abs(o.qpos))) < 1e-7
    finally:
        oracle_py.set_hfield(None)


# ---------------------------------------------------------------------------------------------------------------------------
# CM_FLAG_HFPRISM: one contact per penetrated grid triangle (MuJoCo reports one per penetrated prism: reference
# model/cassie_hfield.xml:4 asks for nconmax = 300 for that reason)
def _prism(model, pc, mc, radius, halflen, margin=0.0, room=32):
    import ctypes
    L = oracle_py.lib()
    L.co_test_hfield_prism.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    out = np.zeros(7 * room)
    pc, mc = np.ascontiguousarray(pc, dtype=float), np.ascontiguousarray(mc, dtype=float)
    n = L.co_test_hfield_prism(ctypes.byref(model.pod), pc.ctypes.data, mc.ctypes.data, radius, halflen, margin, room, out.ctypes.data)
    return n, out.reshape(room, 7)[: min(n, room)]


def _grid():
    nc = 200
    return -5 + 10.0 * np.arange(nc) / (nc - 1)      # sample coordinates of the 200 x 200 grid over [-5, 5]


def test_prism_rule_one_contact_per_penetrated_triangle_geometry(hf):
    """Geometric facts of the definition, on terrains where the answer can be written down:
      - a sphere pressed into flat ground over the INTERIOR of a grid triangle touches that triangle alone: one contact, straight
        down, at the plane model's depth; none when it floats above;
      - over a grid VERTEX all six triangles that meet there are penetrated: six contacts, same depth, vertical normals;
      - a capsule lying along a grid line reports a contact for every triangle within its radius along its whole length, in
        row-major cell order, the (v00, v10, v01) triangle of a cell before the other;
      - over a spike the spike's triangles come out deeper than the flat ones, with normals leaning away from the spike."""
    xs = _grid()
    dx = xs[1] - xs[0]
    eye = np.array([1.0, 0, 0, 0, 1, 0, 0, 0, 1])
    oracle_py.set_hfield(np.zeros((200, 200), dtype=np.float32))
    try:
        j, i = 120, 110
        inside = np.array([xs[j] + dx / 3, xs[i] + dx / 3, -0.1 + 0.018])            # centroid of the lower-left triangle of cell (i, j), 2 mm deep at r = 2 cm
        n, c = _prism(hf, inside, eye, 0.02, 0.0)                                      # (the sphere cuts the ground in a circle of 8.7 mm; the triangle's edges are 11.8 mm away)
        assert n == 1 and abs(c[0, 0] + 0.002) < 1e-12 and np.allclose(c[0, 4:7], [0, 0, 1]) and np.allclose(c[0, 1:3], inside[:2])
        assert abs(c[0, 3] - (-0.1 - 0.001)) < 1e-12                                   # position: midway between the surfaces
        n, _ = _prism(hf, inside + [0, 0, 0.003], eye, 0.02, 0.0)
        assert n == 0
        # pressed 5 mm in, the circle (13.2 mm) reaches over the triangle's edges: the neighbours across them are penetrated too, at
        # their closest points on the shared edges -- shallower, with normals leaning towards the centre (the prism-shaped answer)
        n, c = _prism(hf, inside - [0, 0, 0.003], eye, 0.02, 0.0)
        assert n >= 2 and abs(c[:, 0].min() + 0.005) < 1e-12 and np.sum(np.isclose(c[:, 0], -0.005)) == 1 and np.all(c[:, 6] > 0.7)
        n, c = _prism(hf, [xs[j], xs[i], -0.1 + 0.015], eye, 0.02, 0.0)
        assert n == 6 and np.allclose(c[:, 0], -0.005) and np.allclose(c[:, 4:7], [0, 0, 1])
        # a foot-like capsule (r = 2 cm, half length 8 cm) along x, 1 cm above a grid line y = xs[i]: every triangle whose closest
        # point to the axis is within the radius -- the two rows of cells on either side of the line, both triangles of a cell
        # where the diagonal comes close enough
        along_x = np.array([0.0, 0, 1, 0, 1, 0, -1, 0, 0])                           # capsule axis (third column) = world x
        pc = np.array([xs[j] + 0.5 * dx, xs[i], -0.1 + 0.015])
        n, c = _prism(hf, pc, along_x, 0.02, 0.08)
        # (triangles with an EDGE on the line have a sample straight above that edge: 5 mm; those with only a VERTEX on it take the nearest
        # of the samples, which are 2 cm apart: up to 1 cm off, a little shallower and leaning)
        assert n >= 8 and np.sum(np.isclose(c[:, 0], -0.005)) >= 8 and np.all(c[:, 0] <= -0.004) and np.all(c[:, 0] >= -0.005 - 1e-12) and np.all(c[:, 6] > 0.9)
        assert np.all(np.abs(c[:, 1] - pc[0]) <= 0.08 + 0.02) and np.all(np.abs(c[:, 2] - pc[1]) < 0.01)   # contact points along the axis' foot line
        # the default rule reports the two ends only
        import ctypes
        L = oracle_py.lib()
        L.co_test_hfield_capsule.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        out = np.zeros(28)
        assert L.co_test_hfield_capsule(ctypes.byref(hf.pod), pc.ctypes.data, along_x.ctypes.data, 0.02, 0.08, 0.0, out.ctypes.data) == 2
        # a 5 cm spike at the vertex under the capsule's middle
        h = np.zeros((200, 200), dtype=np.float32)
        h[i, j] = 0.25
        oracle_py.set_hfield(h)
        pc2 = np.array([xs[j], xs[i] + 0.4 * dx, -0.1 + 0.04])
        n, c = _prism(hf, pc2, along_x, 0.02, 0.08)
        assert n >= 2 and c[:, 0].min() < -0.01 and np.any(c[:, 5] > 0.3)              # the spike's flank pushes towards +y
    finally:
        oracle_py.set_hfield(None)


def test_prism_rule_is_the_dense_limit_of_point_samples(hf):
    """Against brute force on the random terrain: for every grid triangle under a capsule, the smallest closest-feature distance
    over the rule's sample spheres, computed here in numpy from the definition's own triangle rule (closest point of the triangle
    when the centre is above its plane, signed plane distance inside its prism otherwise) -- contact count, order, depth."""
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        xs = _grid()
        dx = xs[1] - xs[0]
        rng = np.random.default_rng(3)
        gpos = np.array([0.0, 0.0, -0.1])
        for trial in range(12):
            r, hl = (0.02, 0.08) if trial % 2 else (0.04, 0.21)
            ax = rng.standard_normal(3); ax[2] *= 0.3; ax /= np.linalg.norm(ax)
            b1 = np.cross(ax, [0, 0, 1.0]); b1 /= np.linalg.norm(b1)
            mc = np.stack([b1, np.cross(ax, b1), ax], axis=1).ravel()               # third column = axis
            pc = np.array([rng.uniform(0.5, 1.5), rng.uniform(-1.0, 1.0), -0.1 + rng.uniform(0.05, 0.16)])
            n, c = _prism(hf, pc, mc, r, hl, room=32)
            ns = min(16, 1 + int(np.ceil(2 * hl / r)))
            ts = [hl * (1.0 - 2.0 * k / (ns - 1)) for k in range(ns)]
            p0 = pc - gpos
            reach = r
            j0 = int(np.floor((p0[0] - hl * abs(ax[0]) - reach + 5) / dx)); j1 = int(np.floor((p0[0] + hl * abs(ax[0]) + reach + 5) / dx))
            i0 = int(np.floor((p0[1] - hl * abs(ax[1]) - reach + 5) / dx)); i1 = int(np.floor((p0[1] + hl * abs(ax[1]) + reach + 5) / dx))
            want = []
            for i in range(i0, i1 + 1):
                for j in range(j0, j1 + 1):
                    x0, y0 = xs[j], xs[i]
                    v = {(0, 0): np.array([x0, y0, 0.2 * h[i, j]]), (1, 0): np.array([x0 + dx, y0, 0.2 * h[i, j + 1]]),
                         (0, 1): np.array([x0, y0 + dx, 0.2 * h[i + 1, j]]), (1, 1): np.array([x0 + dx, y0 + dx, 0.2 * h[i + 1, j + 1]])}
                    for tri in ((v[0, 0], v[1, 0], v[0, 1]), (v[1, 1], v[0, 1], v[1, 0])):
                        best = np.inf
                        for t in ts:
                            p = p0 + t * ax
                            a, b_, c_ = tri
                            nrm = np.cross(b_ - a, c_ - a)
                            if nrm[2] < 0:
                                nrm = -nrm
                            nrm /= np.linalg.norm(nrm)
                            s = nrm @ (p - a)
                            if s >= 0:
                                d = np.linalg.norm(p - _closest_on_triangle_np(p, a, b_, c_))
                            else:
                                e = [(q2[0] - q1[0]) * (p[1] - q1[1]) - (q2[1] - q1[1]) * (p[0] - q1[0]) for q1, q2 in ((a, b_), (b_, c_), (c_, a))]
                                d = s if (all(x >= 0 for x in e) or all(x <= 0 for x in e)) else np.inf
                            best = min(best, d)
                        if best - r <= 0:
                            want.append(best - r)
            assert n == len(want), (trial, n, len(want))
            assert np.allclose(c[:, 0], want[:32], atol=1e-12), trial
    finally:
        oracle_py.set_hfield(None)


def test_emulated_kernel_matches_oracle_with_prism_contacts(built):
    """CM_FLAG_HFPRISM in the kernel (hfield_prism_wave: lane = (pair, cell, triangle) key in the oracle's order, the contacts into the
    list by ballot) against the oracle: a robot dropped on the flat patch, one straddling its edge, and one tipped over on the rough
    part, where a lying Cassie needs more rows than one wavefront has lanes -- counts, sweeps and warning bits equal at every step."""
    from cassie_amd import phys as P
    m = _flag_model(P.FLAG_HFPRISM)
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        pod = m.pod
        qs = [m.qpos_init(), m.qpos_init(), m.qpos_init()]
        qs[1][0] = 0.35
        qs[2][0], qs[2][1], qs[2][2] = 0.6, 0.9, 0.75
        qs[2][3:7] = [0.924, 0.0, 0.383, 0.0]
        orcs = [Oracle(pod, q) for q in qs]
        emu = EmuBatch(pod, 3)
        for e, q in enumerate(qs):
            emu.qpos[e] = q
        emu.hfield = h.ravel().copy()
        most, rows = [0, 0, 0], [0, 0, 0]
        for s in range(300):
            emu.step()
            for e, o in enumerate(orcs):
                o.step()
                assert (emu.info[e, 0], emu.info[e, 1], emu.info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (s, e)
                want = (1 if o.d.warn_contact_full else 0) | (2 if o.d.warn_constraint_full else 0)
                assert int(emu.warn[e]) & 3 == want or (int(emu.warn[e]) & 3) | want == int(emu.warn[e]) & 3, (s, e)   # (the kernel's bits are sticky)
                most[e], rows[e] = max(most[e], o.d.ncon), max(rows[e], o.d.nefc)
        assert most[0] >= 6 and rows[2] > 64, (most, rows)     # several triangles under each foot; the lying robot is past one wavefront's rows
        for e, o in enumerate(orcs):
            assert np.max(np.abs(emu.qpos[e] - o.qpos) / np.maximum(1, np.abs(o.qpos))) < 1e-7
    finally:
        oracle_py.set_hfield(None)


def test_caps_follow_the_contact_definition(built):
    """cm_model_t::maxcon / maxefc are set by the model compile with the contact definition: 16 contacts / 63 rows for the default
    definitions -- the stepping launch then has two tiers and no 127-row pass behind it (that pass's workgroups need half a CU each
    and cost config 2 4 % even when empty, profiles/round5/wide_pass_ab.txt) --, 32 / 127 with CM_FLAG_HFPRISM on a model of the
    32-dof Cassie dof tree that has height-field pairs, and back when the flag is cleared."""
    from cassie_amd import phys as P
    for name in ("cassie", "cassie_hfield", "cassie_tray_box"):
        pod = Model(name).pod
        assert (pod.maxcon, pod.maxefc) == (16, 63), name
    m = Model("cassie_hfield")
    m.set_flag(P.FLAG_HFPRISM, True)
    assert (m.pod.maxcon, m.pod.maxefc) == (32, 127)
    m.set_flag(P.FLAG_HFPRISM, False)
    assert (m.pod.maxcon, m.pod.maxefc) == (16, 63)
    plain = Model("cassie")                      # (no height-field pairs: the flag changes nothing)
    plain.set_flag(P.FLAG_HFPRISM, True)
    assert (plain.pod.maxcon, plain.pod.maxefc) == (16, 63)
