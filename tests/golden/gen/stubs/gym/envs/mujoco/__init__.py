from . import mujoco_env  # noqa: F401
