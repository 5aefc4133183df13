"""Import stub so the reference's dp_env_*.py modules can be imported for fixture generation.
mujoco-py / MuJoCo 2.0 are closed-source third-party code that is not installable here; only the
names the reference imports are provided and none of them computes anything."""


class MujocoException(Exception):
    pass


def load_model_from_xml(*a, **k):
    raise RuntimeError("mujoco_py stub: physics is not available in the fixture generator")


load_model_from_path = load_model_from_xml


class MjSim(object):
    def __init__(self, *a, **k):
        raise RuntimeError("mujoco_py stub")


class MjViewer(object):
    def __init__(self, *a, **k):
        raise RuntimeError("mujoco_py stub")
