"""The DeepMimic humanoid of `dp_env_v3.xml`, restated as a Python table (no XML needed at run time).

Source of every number: src/mujoco/humanoid_deepmimic/envs/asset/dp_env_v3.xml (line numbers in
comments).  The same description can be obtained from the XML itself with `mjcf.load_mjcf(path)`;
tests check that the two agree when the reference checkout is present.

Description format (plain dicts/lists, "spec"):
    option : timestep, iterations, tolerance, gravity, solref, solimp
    bodies : list of {name, parent (index, 0 = world), pos}
    joints : list of {name, body, type ('free'|'hinge'), axis, range, limited, armature, damping}
    geoms  : list of {name, body, type ('plane'|'sphere'|'capsule'|'box'), size, pos | fromto, mass,
                      condim, friction, margin, contype, conaffinity}
    motors : list of {name, joint (index), gear, ctrlrange}
    excludes : list of (body, body)
"""

# MuJoCo 2.0 defaults that the XML does not override (EXTERNAL: MuJoCo XML reference, `option`/`geom`)
DEFAULT_OPTION = dict(timestep=0.002, iterations=100, tolerance=1e-8, gravity=(0.0, 0.0, -9.81),
                      solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0), integrator="Euler",
                      solver="Newton", cone="pyramidal")
DEFAULT_GEOM_FRICTION = (1.0, 0.005, 0.0001)


def humanoid_spec():
    opt = dict(DEFAULT_OPTION)
    opt.update(timestep=0.0166, iterations=50, integrator="RK4", solver="PGS")            # :9
    bodies = [dict(name="world", parent=0, pos=(0.0, 0.0, 0.0))]
    joints, geoms, motors = [], [], []

    def body(name, parent, pos):
        bodies.append(dict(name=name, parent=parent, pos=tuple(pos)))
        return len(bodies) - 1

    def hinge(name, b, axis, lo, hi):
        joints.append(dict(name=name, body=b, type="hinge", axis=tuple(axis), range=(lo, hi), limited=True,
                           armature=1.0, damping=1.0))                                       # defaults :4

    def xyz(prefix, b, rx, ry, rz):
        hinge(prefix + "_x", b, (1, 0, 0), *rx)
        hinge(prefix + "_y", b, (0, 1, 0), *ry)
        hinge(prefix + "_z", b, (0, 0, 1), *rz)

    def geom(name, b, gtype, size, mass, pos=None, fromto=None, condim=1, friction=DEFAULT_GEOM_FRICTION):
        geoms.append(dict(name=name, body=b, type=gtype, size=tuple(size), mass=mass,
                          pos=None if pos is None else tuple(pos),
                          fromto=None if fromto is None else tuple(fromto),
                          condim=condim, friction=tuple(friction), margin=0.001,              # defaults :5
                          contype=1, conaffinity=1))

    geom("floor", 0, "plane", (50, 50, 0.2), 0.0, pos=(0, 0, 0), condim=3, friction=(1, 0.1, 0.1))   # :19
    root = body("root", 0, (0, 0, 0.9))                                                            # :21
    geom("root", root, "sphere", (0.09,), 6.0, pos=(0, 0, 0.07))                                   # :22
    joints.append(dict(name="root", body=root, type="free", axis=(0, 0, 1), range=(0, 0), limited=False,
                       armature=0.0, damping=0.0))                                                 # :25
    chest = body("chest", root, (0, 0, 0.236151))                                                  # :28
    geom("chest", chest, "sphere", (0.11,), 14.0, pos=(0, 0, 0.12))                                # :29
    xyz("chest", chest, (-1.2, 1.2), (-1.2, 1.2), (-1.2, 1.2))                                     # :30-32
    neck = body("neck", chest, (0, 0, 0.223894))                                                   # :33
    geom("neck", neck, "sphere", (0.1025,), 2.0, pos=(0, 0, 0.175))                                # :34
    xyz("neck", neck, (-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0))                                       # :35-37
    for side, sy, rx in (("right", -1.0, (-3.14, 0.5)), ("left", 1.0, (-0.5, 3.14))):              # :41-66
        sh = body(side + "_shoulder", chest, (-0.02405, sy * 0.18311, 0.2435))
        geom(side + "_shoulder", sh, "capsule", (0.045,), 1.5, fromto=(0, 0, -0.05, 0, 0, -0.23))
        xyz(side + "_shoulder", sh, rx, (-3.14, 0.7), (-1.5, 1.5))
        el = body(side + "_elbow", sh, (0, 0, -0.274788))
        geom(side + "_elbow", el, "capsule", (0.04,), 1.0, fromto=(0, 0, -0.0525, 0, 0, -0.1875))
        hinge(side + "_elbow", el, (0, -1, 0), 0.0, 2.8)
        geom(side + "_wrist", el, "sphere", (0.04,), 0.5, pos=(0, 0, -0.258947))
    for side, sy in (("right", -1.0), ("left", 1.0)):                                              # :69-106
        hip = body(side + "_hip", root, (0, sy * 0.084887, 0))
        geom(side + "_hip", hip, "capsule", (0.055,), 4.5, fromto=(0, 0, -0.06, 0, 0, -0.36))
        xyz(side + "_hip", hip, (-1.2, 1.2), (-2.57, 1.57), (-1.0, 1.0))
        knee = body(side + "_knee", hip, (0, 0, -0.421546))
        geom(side + "_knee", knee, "capsule", (0.05,), 3.0, fromto=(0, 0, -0.045, 0, 0, -0.355))
        hinge(side + "_knee", knee, (0, -1, 0), -2.7, 0.0)
        ankle = body(side + "_ankle", knee, (0, 0, -0.40987))
        xyz(side + "_ankle", ankle, (-1.0, 1.0), (-1.0, 1.57), (-1.0, 1.0))
        geom(side + "_ankle", ankle, "box", (0.0885, 0.045, 0.0275), 1.0, pos=(0.045, 0, -0.0225))

    names = [b["name"] for b in bodies]
    excludes = [(names.index(a), names.index(b)) for a, b in (                                      # :110-117
        ("right_hip", "root"), ("left_hip", "root"), ("right_hip", "right_knee"), ("left_hip", "left_knee"),
        ("right_knee", "right_ankle"), ("left_knee", "left_ankle"), ("right_elbow", "right_shoulder"),
        ("left_elbow", "left_shoulder"))]
    gears = dict(chest=200, neck=50, right_shoulder=100, right_elbow=60, left_shoulder=100, left_elbow=60,
                 right_hip=200, right_knee=150, right_ankle=90, left_hip=200, left_knee=150, left_ankle=90)
    for ji, j in enumerate(joints):                                                                 # :121-155
        if j["type"] != "hinge":
            continue
        base = j["name"][:-2] if j["name"][-2:] in ("_x", "_y", "_z") else j["name"]
        motors.append(dict(name=j["name"], joint=ji, gear=float(gears[base]), ctrlrange=(-0.5, 0.5)))
    return dict(option=opt, bodies=bodies, joints=joints, geoms=geoms, motors=motors, excludes=excludes)
