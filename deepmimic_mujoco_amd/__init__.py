"""deepmimic_mujoco_amd — MI355X-native batched DeepMimic humanoid environment (the dp_env_v3 hot path).

Host side is pure Python over the C ABI of libdmenv.so (include/dmenv.h, HIP kernels for gfx950):
    DPEnv      the reference's gym-style single environment (src/dp_env_v3.py), one GPU wavefront
    DPVecEnv   N environments in lock step, one wavefront each
    Batch      the raw dm_batch owner (numpy or torch-CUDA buffers)
    MocapDM    DeepMimic clip loader (src/mujoco/mocap_v2.py)
    Config     clip/model selection (src/config.py)
    MlpPolicy  the learner's policy/value network, batched on the env's device (src/mlp_policy_trpo.py)
    traj_segment_generator / add_vtarg_and_adv   device-resident rollouts + GAE (src/trpo.py:27-94)
    load_checkpoint   reader for the reference's tf.train.Saver bundles
    trpo.learn / TrpoLearner   the reference's TRPO learner on torch autograd + RCCL all-mean (src/trpo.py:97-319)
    logio             progress.csv / monitor.csv readers and writers
"""
from .config import Config  # noqa: F401
from .mocap import MocapDM  # noqa: F401
from .model import CompiledModel  # noqa: F401
from .humanoid import humanoid_spec  # noqa: F401
from .mjcf import load_mjcf, to_mjcf  # noqa: F401
from .batch import Batch  # noqa: F401
from .dp_env import DPEnv, DPVecEnv  # noqa: F401
from .tf_checkpoint import load_checkpoint  # noqa: F401
from .policy import MlpPolicy, RunningMeanStd  # noqa: F401
from .trpo import TrpoLearner, learn  # noqa: F401
from . import logio  # noqa: F401
from .rollout import traj_segment_generator, pipelined_segment_generator, SegmentCollector, can_fuse, add_vtarg_and_adv, flatten_segment, RolloutBlock, shard_range  # noqa: F401

__all__ = ["Config", "MocapDM", "CompiledModel", "humanoid_spec", "load_mjcf", "to_mjcf", "Batch", "DPEnv", "DPVecEnv",
           "load_checkpoint", "MlpPolicy", "RunningMeanStd", "traj_segment_generator", "pipelined_segment_generator", "SegmentCollector", "add_vtarg_and_adv", "flatten_segment",
           "RolloutBlock", "shard_range", "TrpoLearner", "learn", "logio"]
