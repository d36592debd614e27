"""Host-side model compiler: hand MJCF + object MJCF -> flat constant tables ("model blob").

What it replaces in the reference
---------------------------------
* the MJCF merge of hand and object (``uhc/data_loaders/mjxml/MujocoXML.py:72-106`` called from
  ``uhc/data_loaders/dataset_singledepth.py:144-150``): object ``<body>`` elements are appended
  to the hand's ``<worldbody>``, so the object is the LAST body and its geoms are the LAST geoms
  (the env hard-codes this, ``uhc/envs/ho_im4.py:77-94``);
* ``mujoco_py.load_model_from_path`` (``uhc/khrylib/rl/envs/common/mujoco_env.py:18-34``), i.e.
  MuJoCo's XML compiler restricted to the features these models use: nested default classes with
  ``childclass``/``class`` resolution, slide/hinge/free joints, box/capsule/plane/mesh geoms,
  explicit ``<inertial>`` or inertia-from-geoms, explicit contact ``<pair>``s, ``<motor>``s.

The output is a self-describing binary blob (named arrays) that both the HIP library
(``hoic_amd/csrc``) and the CPU oracle (``oracle/``) load through ``include/hoic_model.h``.

MuJoCo semantics restated here come from MuJoCo's public documentation (XML reference and
"Computation" chapter) — MuJoCo itself is not in the reference tree; each such
place is marked [MJ-doc].
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# MuJoCo enum values [MJ-doc]
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX, GEOM_MESH = 0, 2, 3, 6, 7
_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
               "box": GEOM_BOX, "mesh": GEOM_MESH}
_JNT_TYPES = {"free": JNT_FREE, "slide": JNT_SLIDE, "hinge": JNT_HINGE}

MINVAL = 1e-15  # mjMINVAL

BLOB_MAGIC = b"HOICMDL1"


# ----------------------------------------------------------------------------- small math
def _fl(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return v


def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat_to_quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0 branch by largest diagonal."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def euler_xyz_to_quat(e):
    """MuJoCo default eulerseq "xyz" (intrinsic x, then y', then z'') [MJ-doc]."""
    q = np.array([1.0, 0, 0, 0])
    for ang, ax in zip(e, range(3)):
        h = 0.5 * ang
        r = np.array([math.cos(h), 0, 0, 0])
        r[1 + ax] = math.sin(h)
        q = quat_mul(q, r)
    return q


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


# ----------------------------------------------------------------------------- STL / mesh
def load_stl(path):
    """Binary or ASCII STL -> (F,3,3) triangle array."""
    data = open(path, "rb").read()
    n = struct.unpack("<I", data[80:84])[0] if len(data) >= 84 else -1
    if n >= 0 and 84 + 50 * n == len(data):
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
        return np.frombuffer(data[84:], dtype=rec)["v"].astype(np.float64)
    verts = []
    for line in data.decode("ascii", "ignore").splitlines():
        t = line.split()
        if len(t) == 4 and t[0] == "vertex":
            verts.append([float(t[1]), float(t[2]), float(t[3])])
    return np.array(verts, dtype=np.float64).reshape(-1, 3, 3)


def mesh_mass_props(tris):
    """Volume, centre of mass and inertia (about the CoM, unit density) of a closed mesh by
    signed tetrahedra from the origin; equals MuJoCo's legacy surface-centroid scheme for the
    convex meshes used here [MJ-doc]."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    vol6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = vol6.sum() / 6.0
    if vol < 0:  # inward-facing winding
        return mesh_mass_props(tris[:, ::-1])
    com = ((a + b + c) / 4.0 * (vol6 / 6.0)[:, None]).sum(0) / vol
    # second moments: integral of x x^T over each tet (origin, a, b, c)
    C = np.zeros((3, 3))
    for t in range(tris.shape[0]):
        A = np.stack([a[t], b[t], c[t]], axis=1)  # columns
        S = A @ (np.ones((3, 3)) + np.eye(3)) @ A.T
        C += vol6[t] / 120.0 * S
    C -= vol * np.outer(com, com)
    inertia = np.trace(C) * np.eye(3) - C
    return vol, com, inertia


def eig3_desc(I):
    """Principal moments in DECREASING order and the rotation whose columns are the axes
    (right-handed) [MJ-doc: mju_eig3 ordering]."""
    w, v = np.linalg.eigh(I)
    order = np.argsort(-w, kind="stable")
    w = w[order]
    v = v[:, order]
    # canonical sign: make the largest-magnitude component of each axis positive, keep det=+1
    for k in range(3):
        j = np.argmax(np.abs(v[:, k]))
        if v[j, k] < 0:
            v[:, k] = -v[:, k]
    if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
    if np.allclose(I, np.diag(np.diag(I)), atol=1e-14 * max(1.0, np.abs(I).max())) and \
            np.all(np.diff(np.diag(I)) <= 0):
        return np.diag(I).copy(), np.eye(3)
    return w, v


# ----------------------------------------------------------------------------- defaults
class _Defaults:
    """Nested <default class=...> tree: attributes of a class = parent's merged with its own."""

    def __init__(self):
        self.classes = {"main": {}}

    def load(self, node, parent="main"):
        for d in node.findall("default"):
            self._load_class(d, None)

    def _load_class(self, d, parent):
        name = d.get("class", "main")
        merged = {k: dict(v) for k, v in (self.classes.get(parent, {}) if parent else {}).items()}
        if name in self.classes and parent is None:
            for k, v in self.classes[name].items():
                merged.setdefault(k, {}).update(v)
        for child in d:
            if child.tag == "default":
                continue
            merged.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = merged
        for sub in d.findall("default"):
            self._load_class(sub, name)

    def attrs(self, tag, elem, childclass):
        cls = elem.get("class") or childclass or "main"
        out = dict(self.classes.get(cls, {}).get(tag, {}))
        out.update(elem.attrib)
        return out


@dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    inertial: dict | None = None
    joints: list = field(default_factory=list)
    geoms: list = field(default_factory=list)


# ----------------------------------------------------------------------------- compiler
class CompiledModel:
    """Named constant tables; ``arrays`` maps name -> ndarray (float64 or int32)."""

    def __init__(self):
        self.arrays: dict[str, np.ndarray] = {}
        self.body_names: list[str] = []
        self.joint_names: list[str] = []
        self.geom_names: list[str] = []
        self.actuator_names: list[str] = []

    def __getattr__(self, k):
        arrays = self.__dict__.get("arrays", {})
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    def scalar(self, k):
        return self.arrays[k].reshape(-1)[0].item()

    # ---- blob (de)serialisation: see include/hoic_model.h for the layout
    def to_blob(self) -> bytes:
        names = sorted(self.arrays)
        entry = struct.Struct("<32siiiiiiqq")
        head = struct.Struct("<8sii")
        off = head.size + entry.size * len(names)
        off = (off + 63) // 64 * 64
        table, chunks = [], []
        for n in names:
            a = np.ascontiguousarray(self.arrays[n])
            if a.dtype == np.float64:
                code = 0
            elif a.dtype == np.int32:
                code = 1
            else:
                raise TypeError(f"{n}: {a.dtype}")
            shape = list(a.shape) + [1] * (4 - a.ndim)
            raw = a.tobytes()
            table.append(entry.pack(n.encode(), code, a.ndim, *shape, off, len(raw)))
            pad = (-len(raw)) % 64
            chunks.append(raw + b"\0" * pad)
            off += len(raw) + pad
        body = head.pack(BLOB_MAGIC, 1, len(names)) + b"".join(table)
        body += b"\0" * ((-len(body)) % 64)
        return body + b"".join(chunks)

    @staticmethod
    def from_blob(blob: bytes) -> "CompiledModel":
        head = struct.Struct("<8sii")
        entry = struct.Struct("<32siiiiiiqq")
        magic, ver, n = head.unpack_from(blob, 0)
        if magic != BLOB_MAGIC or ver != 1:
            raise ValueError("not a HOIC model blob")
        m = CompiledModel()
        for i in range(n):
            name, code, ndim, s0, s1, s2, s3, off, nb = entry.unpack_from(blob, head.size + i * entry.size)
            name = name.rstrip(b"\0").decode()
            dt = np.float64 if code == 0 else np.int32
            m.arrays[name] = np.frombuffer(blob, dtype=dt, count=nb // np.dtype(dt).itemsize,
                                           offset=off).reshape([s0, s1, s2, s3][:ndim]).copy()
        m._names_from_arrays()
        return m

    def _names_from_arrays(self):
        def dec(a):
            return [bytes(r).rstrip(b"\0").decode() for r in a.astype(np.uint8)]
        for attr, key in (("body_names", "names_body"), ("joint_names", "names_joint"),
                          ("geom_names", "names_geom"), ("actuator_names", "names_actuator")):
            if key in self.arrays:
                setattr(self, attr, dec(self.arrays[key]))


def _enc_names(names, width=32):
    out = np.zeros((len(names), width), dtype=np.int32)
    for i, n in enumerate(names):
        b = n.encode()[: width - 1]
        out[i, : len(b)] = list(b)
    return out


def compile_model(hand_xml: str, obj_xml: str | None, max_mesh_verts: int | None = None) -> CompiledModel:
    """Compile hand (+ optional object) MJCF files into constant tables."""
    hand_root = ET.parse(hand_xml).getroot()
    defaults = _Defaults()
    defaults.load(hand_root)
    opt = hand_root.find("option")
    timestep = float(opt.get("timestep", "0.002")) if opt is not None else 0.002
    iterations = int(opt.get("iterations", "100")) if opt is not None else 100

    mesh_assets: dict[str, dict] = {}

    def read_meshes(root, base_dir):
        asset = root.find("asset")
        if asset is None:
            return
        for me in asset.findall("mesh"):
            f = me.get("file")
            scale = _fl(me.get("scale"), 3, [1, 1, 1])
            name = me.get("name") or os.path.splitext(os.path.basename(f))[0]
            mesh_assets[name] = {"file": os.path.join(base_dir, f), "scale": scale}

    read_meshes(hand_root, os.path.dirname(hand_xml))

    bodies: list[_Body] = [_Body("world", 0, np.zeros(3), np.array([1.0, 0, 0, 0]))]

    def walk(elem, parent_id, childclass):
        for ch in elem:
            if ch.tag == "geom":
                bodies[parent_id].geoms.append(defaults.attrs("geom", ch, childclass))
            elif ch.tag == "joint":
                bodies[parent_id].joints.append(defaults.attrs("joint", ch, childclass))
            elif ch.tag == "freejoint":
                a = dict(ch.attrib)
                a["type"] = "free"
                bodies[parent_id].joints.append(a)
            elif ch.tag == "inertial":
                bodies[parent_id].inertial = dict(ch.attrib)
            elif ch.tag == "body":
                cc = ch.get("childclass") or childclass
                q = _fl(ch.get("quat"), 4)
                if q is None:
                    e = _fl(ch.get("euler"), 3)
                    q = euler_xyz_to_quat(e) if e is not None else np.array([1.0, 0, 0, 0])
                q = q / np.linalg.norm(q)
                b = _Body(ch.get("name", f"body{len(bodies)}"), parent_id, _fl(ch.get("pos"), 3, [0, 0, 0]), q)
                bodies.append(b)
                walk(ch, len(bodies) - 1, cc)

    walk(hand_root.find("worldbody"), 0, None)
    if obj_xml is not None:
        obj_root = ET.parse(obj_xml).getroot()
        read_meshes(obj_root, os.path.dirname(obj_xml))
        # reference merge: object bodies appended to the hand's worldbody (MujocoXML.py:89-91)
        walk(obj_root.find("worldbody"), 0, None)

    nbody = len(bodies)
    m = CompiledModel()
    A = m.arrays

    # ---------------- joints / dofs
    jnt_type, jnt_body, jnt_qadr, jnt_dadr, jnt_pos, jnt_axis, jnt_range, jnt_limited = [], [], [], [], [], [], [], []
    jnt_margin, jnt_solref, jnt_solimp, jnt_names = [], [], [], []
    dof_body, dof_jnt, dof_parent, dof_arm, dof_damp, dof_floss, dof_solref, dof_solimp = [], [], [], [], [], [], [], []
    body_jntadr = np.full(nbody, -1, np.int32)
    body_jntnum = np.zeros(nbody, np.int32)
    body_dofadr = np.full(nbody, -1, np.int32)
    body_dofnum = np.zeros(nbody, np.int32)
    qpos0 = []
    nq = nv = 0
    body_lastdof = np.full(nbody, -1, np.int32)  # last dof on the path root->body
    for b, B in enumerate(bodies):
        par_last = body_lastdof[B.parent] if b > 0 else -1
        last = par_last
        if B.joints:
            body_jntadr[b] = len(jnt_type)
            body_dofadr[b] = nv
        for J in B.joints:
            t = _JNT_TYPES[J.get("type", "hinge")]
            jnt_type.append(t)
            jnt_body.append(b)
            jnt_qadr.append(nq)
            jnt_dadr.append(nv)
            jnt_names.append(J.get("name", f"joint{len(jnt_type)}"))
            jnt_pos.append(_fl(J.get("pos"), 3, [0, 0, 0]))
            ax = _fl(J.get("axis"), 3, [0, 0, 1])
            jnt_axis.append(ax / np.linalg.norm(ax))
            jnt_range.append(_fl(J.get("range"), 2, [0, 0]))
            jnt_limited.append(1 if J.get("limited", "false") == "true" else 0)
            jnt_margin.append(float(J.get("margin", "0")))
            jnt_solref.append(_fl(J.get("solreflimit"), 2, [0.02, 1]))
            jnt_solimp.append(_fl(J.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2]))
            nd = 6 if t == JNT_FREE else 1
            for k in range(nd):
                dof_body.append(b)
                dof_jnt.append(len(jnt_type) - 1)
                dof_parent.append(last)
                last = nv + k
                dof_arm.append(float(J.get("armature", "0")))
                dof_damp.append(float(J.get("damping", "0")))
                dof_floss.append(float(J.get("frictionloss", "0")))
                dof_solref.append(_fl(J.get("solreffriction"), 2, [0.02, 1]))
                dof_solimp.append(_fl(J.get("solimpfriction"), 5, [0.9, 0.95, 0.001, 0.5, 2]))
            if t == JNT_FREE:
                qpos0 += list(B.pos) + list(B.quat)
                nq += 7
            else:
                qpos0.append(float(J.get("ref", "0")))
                nq += 1
            nv += nd
        body_jntnum[b] = len(B.joints)
        body_dofnum[b] = nv - body_dofadr[b] if B.joints else 0
        body_lastdof[b] = last
    njnt = len(jnt_type)

    # ---------------- geoms
    g_type, g_body, g_size, g_pos, g_quat, g_contype, g_conaff, g_condim = [], [], [], [], [], [], [], []
    g_fric, g_solref, g_solimp, g_solmix, g_margin, g_gap, g_density, g_mass, g_mesh, g_names = ([] for _ in range(10))
    g_rbound = []
    meshes = []  # per used mesh: dict(vert (n,3) centred hull vertices)
    mesh_index: dict[str, int] = {}

    def get_mesh(name, need_hull):
        key = name
        if key in mesh_index:
            return mesh_index[key]
        ma = mesh_assets[name]
        tris = load_stl(ma["file"]) * ma["scale"][None, None, :]
        vol, com, inertia = mesh_mass_props(tris)
        # MuJoCo re-expresses a mesh in its own inertial frame (centre of mass + principal
        # axes) and folds that offset into the geom pose [MJ-doc].
        pm, R = eig3_desc(inertia)
        entry = {"vol": vol, "com": com, "R": R, "pm": pm, "hull": np.zeros((0, 3)), "planes": np.zeros((0, 4))}
        if need_hull:
            from scipy.spatial import ConvexHull
            pts = np.unique(tris.reshape(-1, 3), axis=0)
            hull = ConvexHull(pts)
            hv = pts[hull.vertices]
            # the reference collides against the whole convex hull of the STL; `max_mesh_verts` (default: no limit)
            # trades that for speed: the hull of the k vertices chosen to minimise the one-sided Hausdorff distance
            entry["hull_full_verts"] = int(hv.shape[0])
            entry["hull_error"] = 0.0
            if max_mesh_verts is not None and hv.shape[0] > max_mesh_verts:
                full = hv
                hv = _decimate_hull(hv, max_mesh_verts)
                entry["hull_error"] = hull_inner_distance(full, hv)
            hv = (hv - com) @ R  # in mesh frame
            # the collision shape is the hull of the (possibly decimated) vertex set: keep its vertices AND its
            # face planes n.x <= d (merged when coplanar), so narrow-phase queries are plain loops over both
            h2 = ConvexHull(hv)
            hv = hv[h2.vertices]
            h2 = ConvexHull(hv)
            eq = h2.equations.copy()                       # n.x + off <= 0 inside
            pl = np.concatenate([eq[:, :3], -eq[:, 3:4]], 1)
            pl = pl[np.lexsort(np.round(pl, 9).T[::-1])]
            keep = [0]
            for i in range(1, pl.shape[0]):
                if np.abs(pl[i] - pl[keep[-1]]).max() > 1e-9:
                    keep.append(i)
            pl = pl[keep]
            # Table order = run order of the simulator's narrow phase: consecutive runs of HULL_RUN_VERTS vertices are
            # spatially compact and consecutive runs of HULL_RUN_FACES faces have similar normals, so that a run's bounding
            # sphere / normal box (built at load time, hoic_capi.hip build_model) lets the kernels skip most runs of a
            # query.  The order is part of the model: "first vertex / first face in table order" tie-breaks of the oracle
            # and of the kernels refer to it.
            hv = hv[coherent_order(hv, HULL_RUN_VERTS)]
            entry["hull"] = hv
            entry["planes"] = pl[coherent_order(pl[:, :3], HULL_RUN_FACES)]
        mesh_index[key] = len(meshes)
        meshes.append(entry)
        return mesh_index[key]

    body_geoms = [[] for _ in range(nbody)]
    for b, B in enumerate(bodies):
        for G in B.geoms:
            t = _GEOM_TYPES[G.get("type", "sphere")]
            gid = len(g_type)
            body_geoms[b].append(gid)
            g_type.append(t)
            g_body.append(b)
            g_names.append(G.get("name", f"geom{gid}"))
            size = _fl(G.get("size"), None, [0, 0, 0])
            size = np.concatenate([size, np.zeros(3 - size.size)]) if size.size < 3 else size[:3]
            pos = _fl(G.get("pos"), 3, [0, 0, 0])
            q = _fl(G.get("quat"), 4)
            if q is None:
                e = _fl(G.get("euler"), 3)
                q = euler_xyz_to_quat(e) if e is not None else np.array([1.0, 0, 0, 0])
            q = q / np.linalg.norm(q)
            contype = int(G.get("contype", "1"))
            conaff = int(G.get("conaffinity", "1"))
            mid = -1
            if t == GEOM_MESH:
                mid = get_mesh(G.get("mesh"), need_hull=(contype != 0 or conaff != 0))
                me = meshes[mid]
                pos = pos + quat_to_mat(q) @ me["com"]
                q = quat_mul(q, mat_to_quat(me["R"]))
                q = q / np.linalg.norm(q)
                hv = me["hull"]
                rb = float(np.linalg.norm(hv, axis=1).max()) if hv.size else 0.0
            elif t == GEOM_BOX:
                rb = float(np.linalg.norm(size))
            elif t == GEOM_CAPSULE:
                rb = float(size[0] + size[1])
            elif t == GEOM_SPHERE:
                rb = float(size[0])
            else:
                rb = 0.0
            g_size.append(size); g_pos.append(pos); g_quat.append(q)
            g_contype.append(contype); g_conaff.append(conaff)
            g_condim.append(int(G.get("condim", "3")))
            g_fric.append(_fl(G.get("friction"), 3, [1, 0.005, 0.0001]))
            g_solref.append(_fl(G.get("solref"), 2, [0.02, 1]))
            g_solimp.append(_fl(G.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2]))
            g_solmix.append(float(G.get("solmix", "1")))
            g_margin.append(float(G.get("margin", "0")))
            g_gap.append(float(G.get("gap", "0")))
            g_density.append(float(G.get("density", "1000")))
            g_mass.append(float(G.get("mass", "-1")))
            g_mesh.append(mid)
            g_rbound.append(rb)
    ngeom = len(g_type)

    # ---------------- body inertial properties
    body_mass = np.zeros(nbody); body_ipos = np.zeros((nbody, 3)); body_iquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    body_inertia = np.zeros((nbody, 3))
    for b, B in enumerate(bodies):
        if B.inertial is not None:
            I = B.inertial
            body_mass[b] = float(I["mass"])
            body_ipos[b] = _fl(I.get("pos"), 3, [0, 0, 0])
            body_inertia[b] = _fl(I.get("diaginertia"), 3)
            iq = _fl(I.get("quat"), 4)
            if iq is None:
                e = _fl(I.get("euler"), 3)
                iq = euler_xyz_to_quat(e) if e is not None else np.array([1.0, 0, 0, 0])
            body_iquat[b] = iq / np.linalg.norm(iq)
        elif b > 0 and body_geoms[b] and B.joints:
            # inertiafromgeom="auto": accumulate geoms [MJ-doc]
            mass = 0.0; com = np.zeros(3); parts = []
            for gid in body_geoms[b]:
                t = g_type[gid]; s = g_size[gid]
                if t == GEOM_BOX:
                    vol = 8 * s[0] * s[1] * s[2]
                    Iu = np.diag([(s[1] ** 2 + s[2] ** 2), (s[0] ** 2 + s[2] ** 2), (s[0] ** 2 + s[1] ** 2)]) / 3.0
                elif t == GEOM_CAPSULE:
                    r, h = s[0], s[1]
                    vc = math.pi * r * r * 2 * h; vs = 4.0 / 3.0 * math.pi * r ** 3
                    vol = vc + vs
                    ixx = vc * (r * r / 4 + h * h / 3) + vs * (2 * r * r / 5 + h * h + 3 * r * h / 4)
                    izz = vc * r * r / 2 + vs * 2 * r * r / 5
                    Iu = np.diag([ixx, ixx, izz]) / vol
                elif t == GEOM_SPHERE:
                    vol = 4.0 / 3.0 * math.pi * s[0] ** 3
                    Iu = np.eye(3) * 0.4 * s[0] ** 2
                elif t == GEOM_MESH:
                    me = meshes[g_mesh[gid]]
                    vol = me["vol"]; Iu = np.diag(me["pm"]) / vol
                else:
                    continue
                gm = g_mass[gid] if g_mass[gid] >= 0 else g_density[gid] * vol
                if gm <= 0:
                    continue
                Rg = quat_to_mat(g_quat[gid])
                parts.append((gm, g_pos[gid], Rg @ (Iu * gm) @ Rg.T))
                mass += gm; com += gm * g_pos[gid]
            if mass > 0:
                com /= mass
                Ifull = np.zeros((3, 3))
                for gm, p, Ig in parts:
                    d = p - com
                    Ifull += Ig + gm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
                pm, R = eig3_desc(Ifull)
                body_mass[b] = mass; body_ipos[b] = com; body_inertia[b] = pm
                body_iquat[b] = mat_to_quat(R)

    # ---------------- actuators (motors, gear 1: qfrc_actuator[dof] = ctrl) [MJ-doc]
    act_dof, act_names = [], []
    act = hand_root.find("actuator")
    if act is not None:
        for a in act:
            j = jnt_names.index(a.get("joint"))
            act_dof.append(jnt_dadr[j]); act_names.append(a.get("name", ""))

    # ---------------- tree bookkeeping
    body_parent = np.array([B.parent for B in bodies], np.int32)
    body_rootid = np.zeros(nbody, np.int32); body_weldid = np.zeros(nbody, np.int32)
    body_depth = np.zeros(nbody, np.int32)
    for b in range(1, nbody):
        p = body_parent[b]
        body_weldid[b] = b if bodies[b].joints else body_weldid[p]
        body_rootid[b] = b if p == 0 else body_rootid[p]
        body_depth[b] = body_depth[p] + 1

    A["nbody"] = np.array([nbody], np.int32); A["njnt"] = np.array([njnt], np.int32)
    A["nq"] = np.array([nq], np.int32); A["nv"] = np.array([nv], np.int32)
    A["nu"] = np.array([len(act_dof)], np.int32); A["ngeom"] = np.array([ngeom], np.int32)
    A["timestep"] = np.array([timestep]); A["iterations"] = np.array([iterations], np.int32)
    A["gravity"] = np.array([0, 0, -9.81]); A["tolerance"] = np.array([1e-8]); A["impratio"] = np.array([1.0])
    A["qpos0"] = np.array(qpos0)
    A["body_parent"] = body_parent; A["body_rootid"] = body_rootid; A["body_weldid"] = body_weldid
    A["body_depth"] = body_depth
    A["body_jntadr"] = body_jntadr; A["body_jntnum"] = body_jntnum
    A["body_dofadr"] = body_dofadr; A["body_dofnum"] = body_dofnum
    A["body_lastdof"] = body_lastdof
    A["body_pos"] = np.array([B.pos for B in bodies]); A["body_quat"] = np.array([B.quat for B in bodies])
    A["body_ipos"] = body_ipos; A["body_iquat"] = body_iquat
    A["body_mass"] = body_mass; A["body_inertia"] = body_inertia
    A["jnt_type"] = np.array(jnt_type, np.int32); A["jnt_bodyid"] = np.array(jnt_body, np.int32)
    A["jnt_qposadr"] = np.array(jnt_qadr, np.int32); A["jnt_dofadr"] = np.array(jnt_dadr, np.int32)
    A["jnt_pos"] = np.array(jnt_pos); A["jnt_axis"] = np.array(jnt_axis)
    A["jnt_range"] = np.array(jnt_range); A["jnt_limited"] = np.array(jnt_limited, np.int32)
    A["jnt_margin"] = np.array(jnt_margin); A["jnt_solref"] = np.array(jnt_solref); A["jnt_solimp"] = np.array(jnt_solimp)
    A["dof_bodyid"] = np.array(dof_body, np.int32); A["dof_jntid"] = np.array(dof_jnt, np.int32)
    A["dof_parentid"] = np.array(dof_parent, np.int32)
    A["dof_armature"] = np.array(dof_arm); A["dof_damping"] = np.array(dof_damp)
    A["dof_frictionloss"] = np.array(dof_floss)
    A["dof_solref"] = np.array(dof_solref); A["dof_solimp"] = np.array(dof_solimp)
    A["geom_type"] = np.array(g_type, np.int32); A["geom_bodyid"] = np.array(g_body, np.int32)
    A["geom_size"] = np.array(g_size); A["geom_pos"] = np.array(g_pos); A["geom_quat"] = np.array(g_quat)
    A["geom_contype"] = np.array(g_contype, np.int32); A["geom_conaffinity"] = np.array(g_conaff, np.int32)
    A["geom_condim"] = np.array(g_condim, np.int32); A["geom_friction"] = np.array(g_fric)
    A["geom_solref"] = np.array(g_solref); A["geom_solimp"] = np.array(g_solimp)
    A["geom_solmix"] = np.array(g_solmix); A["geom_margin"] = np.array(g_margin); A["geom_gap"] = np.array(g_gap)
    A["geom_rbound"] = np.array(g_rbound); A["geom_meshid"] = np.array(g_mesh, np.int32)
    A["act_dofid"] = np.array(act_dof, np.int32)
    # meshes (collision hull vertices, mesh frame)
    vadr, vnum, verts = [], [], []
    for me in meshes:
        vadr.append(sum(vnum)); vnum.append(me["hull"].shape[0]); verts.append(me["hull"])
    A["mesh_vertadr"] = np.array(vadr if vadr else [0], np.int32)
    A["mesh_vertnum"] = np.array(vnum if vnum else [0], np.int32)
    A["mesh_vert"] = np.concatenate(verts, 0) if verts and sum(vnum) else np.zeros((1, 3))
    padr, pnum, planes = [], [], []
    for me in meshes:
        padr.append(sum(pnum)); pnum.append(me["planes"].shape[0]); planes.append(me["planes"])
    A["mesh_planeadr"] = np.array(padr if padr else [0], np.int32)
    A["mesh_planenum"] = np.array(pnum if pnum else [0], np.int32)
    A["mesh_plane"] = np.concatenate(planes, 0) if planes and sum(pnum) else np.zeros((1, 4))
    A["nmesh"] = np.array([len(meshes)], np.int32)
    # provenance of the collision hulls: vertices of the STL's full convex hull and the one-sided Hausdorff distance (m) of
    # the hull in use to it (0 = the full hull, what the reference collides against)
    A["mesh_fullvertnum"] = np.array([me.get("hull_full_verts", me["hull"].shape[0]) for me in meshes] or [0], np.int32)
    A["mesh_hull_error"] = np.array([me.get("hull_error", 0.0) for me in meshes] or [0.0], np.float64)
    A["names_body"] = _enc_names([B.name for B in bodies]); A["names_joint"] = _enc_names(jnt_names)
    A["names_geom"] = _enc_names(g_names); A["names_actuator"] = _enc_names(act_names)
    m._names_from_arrays()

    # ---------------- collision pair list with mixed contact parameters [MJ-doc]
    pairs = []

    def fric5(f3):
        return np.array([f3[0], f3[0], f3[1], f3[2], f3[2]])

    def add_pair(g1, g2, condim, friction, solref, solimp, margin, gap):
        if g_type[g1] > g_type[g2]:
            g1, g2 = g2, g1  # collision table is upper-triangular in geom type
        pairs.append((g1, g2, condim, friction, solref, solimp, margin, gap))

    for b1 in range(nbody):
        for b2 in range(b1 + 1, nbody):
            if body_weldid[b1] == body_weldid[b2]:
                continue  # same rigid assembly (incl. static x static)
            # parent-child filter: welded parents; the world body is exempt
            w1, w2 = body_weldid[b1], body_weldid[b2]
            pw1 = body_weldid[body_parent[w1]] if w1 else 0
            pw2 = body_weldid[body_parent[w2]] if w2 else 0
            if (w1 != 0 and w2 != 0) and (pw1 == w2 or pw2 == w1):
                continue
            for g1 in body_geoms[b1]:
                for g2 in body_geoms[b2]:
                    if not ((g_contype[g1] & g_conaff[g2]) or (g_contype[g2] & g_conaff[g1])):
                        continue
                    condim = max(g_condim[g1], g_condim[g2])
                    fr = np.maximum(g_fric[g1], g_fric[g2])
                    mix = g_solmix[g1] / (g_solmix[g1] + g_solmix[g2])
                    r1, r2 = g_solref[g1], g_solref[g2]
                    if r1[0] > 0 and r2[0] > 0:
                        sr = mix * r1 + (1 - mix) * r2
                    else:
                        sr = np.minimum(r1, r2)
                    si = mix * g_solimp[g1] + (1 - mix) * g_solimp[g2]
                    add_pair(g1, g2, condim, fric5(fr), sr, si,
                             max(g_margin[g1], g_margin[g2]), max(g_gap[g1], g_gap[g2]))
    n_dynamic = len(pairs)
    con = hand_root.find("contact")
    seen = set()
    if con is not None:
        for p in con.findall("pair"):
            g1 = g_names.index(p.get("geom1")); g2 = g_names.index(p.get("geom2"))
            key = (min(g1, g2), max(g1, g2))
            if key in seen:  # duplicate <pair> (hand xml :48 / :52) keeps one contact source
                continue
            seen.add(key)
            add_pair(g1, g2, int(p.get("condim", "3")),
                     _fl(p.get("friction"), 5, [1, 1, 0.005, 0.0001, 0.0001]),
                     _fl(p.get("solref"), 2, [0.02, 1]),
                     _fl(p.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2]),
                     float(p.get("margin", "0")), float(p.get("gap", "0")))
    A["npair"] = np.array([len(pairs)], np.int32); A["npair_dynamic"] = np.array([n_dynamic], np.int32)
    A["pair_geom1"] = np.array([p[0] for p in pairs], np.int32)
    A["pair_geom2"] = np.array([p[1] for p in pairs], np.int32)
    A["pair_condim"] = np.array([p[2] for p in pairs], np.int32)
    A["pair_friction"] = np.array([p[3] for p in pairs]); A["pair_solref"] = np.array([p[4] for p in pairs])
    A["pair_solimp"] = np.array([p[5] for p in pairs])
    A["pair_margin"] = np.array([p[6] for p in pairs]); A["pair_gap"] = np.array([p[7] for p in pairs])

    # ---------------- env-glue indices the reference derives from names (ho_im4.py:74-97)
    hand_body_idx = [i for i, B in enumerate(bodies) if B.name.startswith("link")]
    hg = [i for i, n in enumerate(g_names) if n.startswith("robot0:")]
    og0 = hg[-1] + 1 if hg else 0
    og1 = og0 - 1
    for i in range(og0, ngeom):
        if not g_names[i].startswith("C_"):
            break
        og1 = i
    A["hand_body0"] = np.array([hand_body_idx[0]], np.int32)
    A["hand_nbody"] = np.array([len(hand_body_idx)], np.int32)
    A["obj_body"] = np.array([hand_body_idx[-1] + 1 if obj_xml else -1], np.int32)
    A["hand_geom0"] = np.array([hg[0]], np.int32); A["hand_geom1"] = np.array([hg[-1]], np.int32)
    A["obj_geom0"] = np.array([og0], np.int32); A["obj_geom1"] = np.array([og1], np.int32)
    A["hand_nq"] = np.array([nq - (7 if obj_xml else 0)], np.int32)
    A["hand_nv"] = np.array([nv - (6 if obj_xml else 0)], np.int32)
    A["hand_mass"] = np.array([body_mass[hand_body_idx].sum()])

    _set_const(m)
    return m


def hull_inner_distance(full_pts, sub_pts):
    """One-sided Hausdorff distance from the hull of `full_pts` to the hull of `sub_pts` (a subset, so the second hull
    lies inside the first): the largest distance from a vertex of the full hull to the inner hull, measured as the
    largest plane excess over the inner hull's faces (a lower bound of the Euclidean point-to-polytope distance that is
    exact where the closest point lies on a face, and never above it)."""
    from scipy.spatial import ConvexHull
    eq = ConvexHull(sub_pts).equations
    return float(np.maximum((np.asarray(full_pts) @ eq[:, :3].T + eq[:, 3]).max(axis=1), 0.0).max())


HULL_RUN_VERTS, HULL_RUN_FACES = 64, 32      # = HOIC_HULL_RUN_VERTS / HOIC_HULL_RUN_FACES of include/hoic_model.h


def coherent_order(P, leaf):
    """Permutation of the rows of P (n, d) such that consecutive runs of `leaf` rows are compact: recursive bisection along
    the axis of largest extent, the cut placed at a multiple of `leaf` nearest the middle (so every run but the last
    holds rows of one leaf); stable sorts: deterministic."""
    P = np.asarray(P, dtype=np.float64)
    out = []

    def rec(ids):
        if len(ids) <= leaf:
            out.extend(ids.tolist())
            return
        sub = P[ids]
        ax = int(np.argmax(sub.max(0) - sub.min(0)))
        order = ids[np.argsort(sub[:, ax], kind="stable")]
        k = leaf * max(1, int(round(len(ids) / (2.0 * leaf))))
        rec(order[:k]); rec(order[k:])
    rec(np.arange(P.shape[0]))
    return np.array(out, dtype=np.int64)


def _decimate_hull(hv, k):
    """k hull vertices whose hull approximates the full hull from inside: start from the axis extremes, then repeatedly add
    the vertex that sticks out farthest from the current inner hull (greedy minimisation of the Hausdorff distance; 3-5x
    smaller error than a farthest-point subset at equal k)."""
    from scipy.spatial import ConvexHull
    idx = []
    for a in range(3):
        for j in (int(np.argmax(hv[:, a])), int(np.argmin(hv[:, a]))):
            if j not in idx:
                idx.append(j)
    while len(idx) < k:
        eq = ConvexHull(hv[idx]).equations
        d = (hv @ eq[:, :3].T + eq[:, 3]).max(axis=1)
        j = int(np.argmax(d))
        if d[j] < 1e-12:
            break
        idx.append(j)
    return hv[sorted(idx)]


# ----------------------------------------------------------------------------- qpos0 constants
def fk_numpy(m: CompiledModel, qpos):
    """Forward kinematics [MJ-doc: per body, parent frame * body offset, then joints in order]."""
    A = m.arrays
    nb = m.scalar("nbody")
    xpos = np.zeros((nb, 3)); xquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    nj = m.scalar("njnt")
    xanchor = np.zeros((nj, 3)); xaxis = np.zeros((nj, 3))
    for b in range(1, nb):
        p = A["body_parent"][b]
        ja, jn = A["body_jntadr"][b], A["body_jntnum"][b]
        if jn == 1 and A["jnt_type"][ja] == JNT_FREE:
            qa = A["jnt_qposadr"][ja]
            pos = np.array(qpos[qa:qa + 3]); quat = np.array(qpos[qa + 3:qa + 7]); quat /= np.linalg.norm(quat)
            xanchor[ja] = pos; xaxis[ja] = quat_to_mat(quat)[:, 2]
        else:
            pos = xpos[p] + quat_to_mat(xquat[p]) @ A["body_pos"][b]
            quat = quat_mul(xquat[p], A["body_quat"][b])
            for j in range(ja, ja + jn):
                R = quat_to_mat(quat)
                xanchor[j] = pos + R @ A["jnt_pos"][j]; xaxis[j] = R @ A["jnt_axis"][j]
                q = qpos[A["jnt_qposadr"][j]] - A["qpos0"][A["jnt_qposadr"][j]]
                if A["jnt_type"][j] == JNT_SLIDE:
                    pos = pos + xaxis[j] * q
                else:
                    ax = A["jnt_axis"][j]
                    ql = np.concatenate([[math.cos(q / 2)], math.sin(q / 2) * ax])
                    quat = quat_mul(quat, ql)
                    pos = xanchor[j] - quat_to_mat(quat) @ A["jnt_pos"][j]
            quat /= np.linalg.norm(quat)
        xpos[b] = pos; xquat[b] = quat
    return xpos, xquat, xanchor, xaxis


def dof_subspaces(m, qpos, xpos, xquat, xanchor, xaxis):
    """Spatial motion axes about the world origin: S = [w; v_O]."""
    A = m.arrays
    nv = m.scalar("nv")
    S = np.zeros((nv, 6))
    for j in range(m.scalar("njnt")):
        d = A["jnt_dofadr"][j]; t = A["jnt_type"][j]
        if t == JNT_SLIDE:
            S[d, 3:] = xaxis[j]
        elif t == JNT_HINGE:
            S[d, :3] = xaxis[j]; S[d, 3:] = np.cross(xanchor[j], xaxis[j])
        else:
            b = A["jnt_bodyid"][j]; R = quat_to_mat(xquat[b])
            for k in range(3):
                S[d + k, 3 + k] = 1.0
                S[d + 3 + k, :3] = R[:, k]; S[d + 3 + k, 3:] = np.cross(xpos[b], R[:, k])
    return S


def mass_matrix_numpy(m, qpos):
    A = m.arrays
    nb, nv = m.scalar("nbody"), m.scalar("nv")
    xpos, xquat, xanchor, xaxis = fk_numpy(m, qpos)
    S = dof_subspaces(m, qpos, xpos, xquat, xanchor, xaxis)
    Ic = np.zeros((nb, 6, 6))
    xipos = np.zeros((nb, 3))
    for b in range(1, nb):
        R = quat_to_mat(xquat[b]); c = xpos[b] + R @ A["body_ipos"][b]; xipos[b] = c
        Ri = R @ quat_to_mat(A["body_iquat"][b]); I3 = Ri @ np.diag(A["body_inertia"][b]) @ Ri.T
        mass = A["body_mass"][b]; cx = _skew(c)
        Ic[b, :3, :3] = I3 + mass * cx @ cx.T; Ic[b, :3, 3:] = mass * cx
        Ic[b, 3:, :3] = mass * cx.T; Ic[b, 3:, 3:] = mass * np.eye(3)
    for b in range(nb - 1, 0, -1):
        Ic[A["body_parent"][b]] += Ic[b]
    M = np.zeros((nv, nv))
    for i in range(nv):
        f = Ic[A["dof_bodyid"][i]] @ S[i]
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = S[j] @ f
            j = A["dof_parentid"][j]
        M[i, i] += A["dof_armature"][i]
    return M, S, xpos, xquat, xipos


def _set_const(m: CompiledModel):
    """dof_invweight0 / body_invweight0 / meaninertia at qpos0 [MJ-doc: mj_setConst]."""
    A = m.arrays
    nb, nv = m.scalar("nbody"), m.scalar("nv")
    M, S, xpos, xquat, xipos = mass_matrix_numpy(m, A["qpos0"])
    Minv = np.linalg.inv(M)
    dinv = np.diag(Minv).copy()
    for j in range(m.scalar("njnt")):
        if A["jnt_type"][j] == JNT_FREE:
            d = A["jnt_dofadr"][j]
            dinv[d:d + 3] = dinv[d:d + 3].mean(); dinv[d + 3:d + 6] = dinv[d + 3:d + 6].mean()
    A["dof_invweight0"] = np.maximum(dinv, MINVAL)
    biw = np.zeros((nb, 2))
    for b in range(1, nb):
        if A["body_weldid"][b] == 0:
            continue
        Jp = np.zeros((3, nv)); Jr = np.zeros((3, nv))
        d = A["body_lastdof"][b]
        while d >= 0:
            Jr[:, d] = S[d, :3]; Jp[:, d] = np.cross(S[d, :3], xipos[b]) + S[d, 3:]
            d = A["dof_parentid"][d]
        biw[b, 0] = np.trace(Jp @ Minv @ Jp.T) / 3.0
        biw[b, 1] = np.trace(Jr @ Minv @ Jr.T) / 3.0
    A["body_invweight0"] = np.where(A["body_weldid"][:, None] == 0, 0.0, np.maximum(biw, MINVAL))
    A["meaninertia"] = np.array([np.trace(M) / nv])


# ----------------------------------------------------------------------------- convenience
def compile_reference_config(ref_root: str, obj: str) -> CompiledModel:
    """Compile one of the three release configs straight from a checkout of the reference's
    assets (``config/release/<obj>_future5_light_add_geom.yml``: mujoco_model + obj_fn)."""
    hand = os.path.join(ref_root, "assets/hand_model/spheremesh/sphere_mesh_hand_add_geom.xml")
    objx = os.path.join(ref_root, f"assets/SingleDepth/{obj}_light.xml")
    return compile_model(hand, objx)


def packaged_model_path(obj: str) -> str:
    return os.path.join(os.path.dirname(__file__), "data", f"{obj}.hoicmodel")


def load_packaged(obj: str = "box") -> CompiledModel:
    with open(packaged_model_path(obj), "rb") as f:
        return CompiledModel.from_blob(f.read())
