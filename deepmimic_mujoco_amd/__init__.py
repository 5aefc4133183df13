"""deepmimic_mujoco_amd — MI355X-native batched DeepMimic humanoid environment (the dp_env_v3 hot path).

Host side is pure Python over the C ABI of libdmenv.so (include/dmenv.h, HIP kernels for gfx950):
    DPEnv      the reference's gym-style single environment (src/dp_env_v3.py), one GPU wavefront
    DPVecEnv   N environments in lock step, one wavefront each
    Batch      the raw dm_batch owner (numpy or torch-CUDA buffers)
    MocapDM    DeepMimic clip loader (src/mujoco/mocap_v2.py)
    Config     clip/model selection (src/config.py)
"""
from .config import Config  # noqa: F401
from .mocap import MocapDM  # noqa: F401
from .model import CompiledModel  # noqa: F401
from .humanoid import humanoid_spec  # noqa: F401
from .mjcf import load_mjcf, to_mjcf  # noqa: F401
from .batch import Batch  # noqa: F401
from .dp_env import DPEnv, DPVecEnv  # noqa: F401

__all__ = ["Config", "MocapDM", "CompiledModel", "humanoid_spec", "load_mjcf", "to_mjcf", "Batch", "DPEnv", "DPVecEnv"]
