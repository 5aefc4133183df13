"""Reader for the MJCF subset used by the DeepMimic humanoid models (`dp_env_v3.xml` and its rig variants).

Produces the same "spec" dict as `humanoid.humanoid_spec()`.  Supported: <compiler angle= inertiafromgeom=>,
one top-level <default> with <joint>/<geom>/<motor>, <option>, nested <body> with <joint type=free|hinge>
and <geom type=plane|sphere|capsule|box> (pos/size/fromto/mass/condim/friction/margin/contype/conaffinity),
<contact><exclude>, <actuator><motor gear= joint=>.  Visual-only elements (site, camera, light, asset,
rgba, material) are ignored.  Anything else that would change the physics raises ValueError instead of
being silently dropped.
"""
import math
import xml.etree.ElementTree as ET

from .humanoid import DEFAULT_GEOM_FRICTION, DEFAULT_OPTION

_IGNORED = {"site", "camera", "light", "inertial_visual"}


def _floats(s, n=None):
    v = tuple(float(x) for x in s.split())
    if n is not None and len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def _bool(s):
    return str(s).strip().lower() == "true"


def load_mjcf(path_or_string):
    if path_or_string.lstrip().startswith("<"):
        root = ET.fromstring(path_or_string)
    else:
        root = ET.parse(path_or_string).getroot()
    if root.tag != "mujoco":
        raise ValueError("not an MJCF document")
    comp = root.find("compiler")
    angle_scale = 1.0
    if comp is not None:
        if comp.get("angle", "degree") != "radian":
            angle_scale = math.pi / 180.0
        if not _bool(comp.get("inertiafromgeom", "auto")) and comp.get("inertiafromgeom", "auto") != "auto":
            raise ValueError("only inertiafromgeom=true/auto models are supported")
    else:
        angle_scale = math.pi / 180.0

    dj, dg, dm = {}, {}, {}
    for d in root.findall("default"):
        for child in d:
            if child.tag == "joint":
                dj.update(child.attrib)
            elif child.tag == "geom":
                dg.update(child.attrib)
            elif child.tag == "motor":
                dm.update(child.attrib)
            elif child.tag == "default":
                raise ValueError("nested default classes are not supported")

    opt = dict(DEFAULT_OPTION)
    o = root.find("option")
    if o is not None:
        for k, v in o.attrib.items():
            if k in ("timestep", "tolerance"):
                opt[k] = float(v)
            elif k == "iterations":
                opt[k] = int(v)
            elif k == "gravity":
                opt[k] = _floats(v, 3)
            elif k in ("integrator", "solver", "cone"):
                opt[k] = v
            else:
                raise ValueError("unsupported <option> attribute %r" % k)

    bodies = [dict(name="world", parent=0, pos=(0.0, 0.0, 0.0))]
    joints, geoms = [], []

    def add_geom(e, b):
        a = dict(dg)
        a.update(e.attrib)
        gtype = a.get("type", "sphere")
        if gtype not in ("plane", "sphere", "capsule", "box"):
            raise ValueError("unsupported geom type %r" % gtype)
        for bad in ("quat", "euler", "axisangle", "xyaxes", "zaxis", "density", "solref", "solimp", "gap"):
            if bad in a:
                raise ValueError("unsupported geom attribute %r" % bad)
        geoms.append(dict(
            name=a.get("name", "geom%d" % len(geoms)), body=b, type=gtype, size=_floats(a.get("size", "0")),
            mass=float(a["mass"]) if "mass" in a else (0.0 if gtype == "plane" else None),
            pos=None if "fromto" in a else _floats(a.get("pos", "0 0 0"), 3),
            fromto=_floats(a["fromto"], 6) if "fromto" in a else None,
            condim=int(a.get("condim", 3)),
            friction=(_floats(a["friction"]) + DEFAULT_GEOM_FRICTION[len(_floats(a["friction"])):]) if "friction" in a else DEFAULT_GEOM_FRICTION,
            margin=float(a.get("margin", 0.0)), contype=int(a.get("contype", 1)),
            conaffinity=int(a.get("conaffinity", 1))))
        if geoms[-1]["mass"] is None:
            raise ValueError("geom %r: only explicit mass= is supported" % geoms[-1]["name"])

    def add_joint(e, b):
        a = dict(dj)
        a.update(e.attrib)
        jtype = a.get("type", "hinge")
        if jtype not in ("free", "hinge"):
            raise ValueError("unsupported joint type %r" % jtype)
        if jtype == "free":
            joints.append(dict(name=a.get("name", "joint%d" % len(joints)), body=b, type="free", axis=(0.0, 0.0, 1.0),
                               range=(0.0, 0.0), limited=False, armature=0.0, damping=0.0))
            return
        if any(abs(x) > 0 for x in _floats(a.get("pos", "0 0 0"), 3)):
            raise ValueError("joint pos offsets are not supported")
        if float(a.get("stiffness", 0)) != 0 or float(a.get("frictionloss", 0)) != 0:
            raise ValueError("joint stiffness / frictionloss are not supported")
        rng = _floats(a.get("range", "0 0"), 2)
        joints.append(dict(name=a.get("name", "joint%d" % len(joints)), body=b, type="hinge",
                           axis=_floats(a.get("axis", "0 0 1"), 3),
                           range=(rng[0] * angle_scale, rng[1] * angle_scale),
                           limited=_bool(a.get("limited", "false")),
                           armature=float(a.get("armature", 0)), damping=float(a.get("damping", 0))))

    def walk(elem, parent):
        for e in elem:
            if e.tag == "geom":
                add_geom(e, parent)
            elif e.tag == "joint":
                add_joint(e, parent)
            elif e.tag == "body":
                for bad in ("quat", "euler", "axisangle", "xyaxes", "zaxis"):
                    if bad in e.attrib:
                        raise ValueError("rotated body frames are not supported")
                bodies.append(dict(name=e.get("name", "body%d" % len(bodies)), parent=parent,
                                   pos=_floats(e.get("pos", "0 0 0"), 3)))
                walk(e, len(bodies) - 1)
            elif e.tag in _IGNORED:
                continue
            elif e.tag == "inertial":
                raise ValueError("explicit <inertial> is not supported")
            else:
                raise ValueError("unsupported element <%s> in <body>" % e.tag)

    wb = root.find("worldbody")
    if wb is None:
        raise ValueError("no <worldbody>")
    walk(wb, 0)
    # a body's joints must be contiguous and bodies numbered depth-first, which walk() guarantees; but geoms and
    # joints were appended in document order across bodies — sort joints by body (stable) as MuJoCo numbers them
    joints.sort(key=lambda j: j["body"])
    geoms.sort(key=lambda g: g["body"])

    names = [b["name"] for b in bodies]
    excludes = []
    c = root.find("contact")
    if c is not None:
        for e in c:
            if e.tag != "exclude":
                raise ValueError("unsupported <contact> element <%s>" % e.tag)
            excludes.append((names.index(e.get("body1")), names.index(e.get("body2"))))
    for tag in ("equality", "tendon", "sensor", "keyframe"):
        if root.find(tag) is not None:
            raise ValueError("<%s> is not supported" % tag)

    motors = []
    act = root.find("actuator")
    jnames = [j["name"] for j in joints]
    if act is not None:
        for e in act:
            if e.tag != "motor":
                raise ValueError("unsupported actuator <%s>" % e.tag)
            a = dict(dm)
            a.update(e.attrib)
            cr = _floats(a.get("ctrlrange", "0 0"), 2)
            if not _bool(a.get("ctrllimited", "false")):
                cr = (-float("inf"), float("inf"))
            gear = _floats(a.get("gear", "1"))
            motors.append(dict(name=a.get("name", "motor%d" % len(motors)), joint=jnames.index(a["joint"]),
                               gear=gear[0], ctrlrange=cr))
    return dict(option=opt, bodies=bodies, joints=joints, geoms=geoms, motors=motors, excludes=excludes)


def to_mjcf(spec):
    """Serialise a spec back to MJCF text (used by tests and to hand the model to other tools)."""
    o = spec["option"]
    out = ['<mujoco model="humanoid">', '  <compiler angle="radian" inertiafromgeom="true"/>',
           '  <option integrator="%s" solver="%s" iterations="%d" timestep="%r" tolerance="%r" gravity="%r %r %r"/>' % (
               o["integrator"], o["solver"], o["iterations"], o["timestep"], o["tolerance"], *o["gravity"]),
           "  <worldbody>"]
    children = {}
    for i, b in enumerate(spec["bodies"]):
        if i:
            children.setdefault(b["parent"], []).append(i)

    def fmt(v):
        return " ".join(repr(float(x)) for x in v)

    def emit(bi, ind):
        for g in spec["geoms"]:
            if g["body"] != bi:
                continue
            place = 'fromto="%s"' % fmt(g["fromto"]) if g["fromto"] is not None else 'pos="%s"' % fmt(g["pos"])
            out.append('%s<geom name="%s" type="%s" size="%s" %s mass="%r" condim="%d" friction="%s" margin="%r" '
                       'contype="%d" conaffinity="%d"/>' % (ind, g["name"], g["type"], fmt(g["size"]), place, g["mass"],
                                                            g["condim"], fmt(g["friction"]), g["margin"], g["contype"],
                                                            g["conaffinity"]))
        for j in spec["joints"]:
            if j["body"] != bi:
                continue
            if j["type"] == "free":
                out.append('%s<joint name="%s" type="free"/>' % (ind, j["name"]))
            else:
                out.append('%s<joint name="%s" type="hinge" axis="%s" range="%s" limited="%s" armature="%r" damping="%r"/>' % (
                    ind, j["name"], fmt(j["axis"]), fmt(j["range"]), "true" if j["limited"] else "false",
                    j["armature"], j["damping"]))
        for ci in children.get(bi, []):
            b = spec["bodies"][ci]
            out.append('%s<body name="%s" pos="%s">' % (ind, b["name"], fmt(b["pos"])))
            emit(ci, ind + "  ")
            out.append("%s</body>" % ind)

    emit(0, "    ")
    out.append("  </worldbody>")
    out.append("  <contact>")
    for a, b in spec["excludes"]:
        out.append('    <exclude body1="%s" body2="%s"/>' % (spec["bodies"][a]["name"], spec["bodies"][b]["name"]))
    out.append("  </contact>")
    out.append("  <actuator>")
    for m in spec["motors"]:
        out.append('    <motor name="%s" joint="%s" gear="%r" ctrllimited="true" ctrlrange="%s"/>' % (
            m["name"], spec["joints"][m["joint"]]["name"], m["gear"], fmt(m["ctrlrange"])))
    out.append("  </actuator>")
    out.append("</mujoco>")
    return "\n".join(out)
