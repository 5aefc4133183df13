"""DeepMimic mocap clip loader: the host-side mirror of `MocapDM` (src/mujoco/mocap_v2.py:14-149).

Same public surface (`load_mocap`, `read_raw_data`, `convert_raw_data`, attributes `dt`, `durations`,
`all_states`, `data`, `data_config`, `data_vel`), same numbers (pinned against tests/golden/mocap_*.npz,
which were produced by executing the reference loader), but computed for all frames at once and
ending in three dense float64 tables that are uploaded to the GPU once:

    data        [F, 44]  dura | root_pos(3) | root_quat(4, wxyz) | joints in MuJoCo order (quat / angle)
    data_config [F, 35]  a MuJoCo qpos: root pos, root quat, Euler-rxyz hinge triples / single hinges
    data_vel    [F, 34]  a MuJoCo qvel by finite differences (with the reference's sign quirks kept)

A clip is either a DeepMimic JSON file ({"Loop":..., "Frames": [[44 floats] ...]}) or the name of a
clip in the bundled asset pack (assets/motions.npz: the raw `Frames` arrays of the 15 clips).
"""
import json
import os

import numpy as np

from . import rotations as R

# joint tables (src/mujoco/mocap_util.py:5-29)
BODY_JOINTS = ["chest", "neck", "right_shoulder", "right_elbow", "left_shoulder", "left_elbow",
               "right_hip", "right_knee", "right_ankle", "left_hip", "left_knee", "left_ankle"]
BODY_JOINTS_IN_DP_ORDER = ["chest", "neck", "right_hip", "right_knee", "right_ankle", "right_shoulder",
                           "right_elbow", "left_hip", "left_knee", "left_ankle", "left_shoulder",
                           "left_elbow"]
DOF_DEF = {"root": 3, "chest": 3, "neck": 3, "right_shoulder": 3, "right_elbow": 1, "right_wrist": 0,
           "left_shoulder": 3, "left_elbow": 1, "left_wrist": 0, "right_hip": 3, "right_knee": 1,
           "right_ankle": 3, "left_hip": 3, "left_knee": 1, "left_ankle": 3}
BODY_DEFS = ["root", "chest", "neck", "right_hip", "right_knee", "right_ankle", "right_shoulder",
             "right_elbow", "right_wrist", "left_hip", "left_knee", "left_ankle", "left_shoulder",
             "left_elbow", "left_wrist"]
PARAMS_KP_KD = {"chest": [1000, 100], "neck": [100, 10], "right_shoulder": [400, 40],
                "right_elbow": [300, 30], "left_shoulder": [400, 40], "left_elbow": [300, 30],
                "right_hip": [500, 50], "right_knee": [500, 50], "right_ankle": [400, 40],
                "left_hip": [500, 50], "left_knee": [500, 50], "left_ankle": [400, 40]}
JOINT_WEIGHT = {"root": 1, "chest": 0.5, "neck": 0.3, "right_hip": 0.5, "right_knee": 0.3,
                "right_ankle": 0.2, "right_shoulder": 0.3, "right_elbow": 0.2, "right_wrist": 0.0,
                "left_hip": 0.5, "left_knee": 0.3, "left_ankle": 0.2, "left_shoulder": 0.3,
                "left_elbow": 0.2, "left_wrist": 0.0}

ALL_CLIPS = ["backflip", "cartwheel", "crawl", "dance_a", "dance_b", "getup_facedown", "getup_faceup",
             "jump", "kick", "punch", "roll", "run", "spin", "spinkick", "walk"]

_ASSET_PACK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "motions.npz")


def _raw_column_layout():
    """Column ranges of one raw frame, in DeepMimic file order (src/mujoco/mocap_v2.py:44-58)."""
    cols = {}
    off = 8
    for j in BODY_JOINTS_IN_DP_ORDER:
        n = 1 if DOF_DEF[j] == 1 else 4
        cols[j] = (off, off + n)
        off += n
    assert off == 44
    return cols


def load_frames(clip_or_path):
    """Return (frames[F,44] float64, loop:str) for a clip name or a DeepMimic JSON file."""
    if os.path.isfile(clip_or_path):
        with open(clip_or_path, "r") as fin:
            js = json.load(fin)
        frames = np.array(js["Frames"], dtype=np.float64)
        loop = str(js.get("Loop", "none"))
    else:
        name = os.path.basename(clip_or_path)
        if name.startswith("humanoid3d_"):
            name = name[len("humanoid3d_"):]
        if name.endswith(".txt"):
            name = name[:-4]
        if name not in ALL_CLIPS:
            raise FileNotFoundError("no mocap file %r and no bundled clip named %r" % (clip_or_path, name))
        with np.load(_ASSET_PACK) as pack:
            frames = np.array(pack["frames_" + name], dtype=np.float64)
            loop = str(pack["loop_" + name])
    if frames.ndim != 2 or frames.shape[1] != 44:
        raise ValueError("mocap Frames must be [F, 44], got %r" % (frames.shape,))
    return frames, loop


class MocapDM(object):
    def __init__(self):
        self.num_bodies = len(BODY_DEFS)
        self.pos_dim = 3
        self.rot_dim = 4
        self.loop = "none"

    def load_mocap(self, filepath):
        self.read_raw_data(filepath)
        self.convert_raw_data()

    # -- stage 1: raw frames -> aligned per-joint states (src/mujoco/mocap_v2.py:24-62) -------------
    def read_raw_data(self, filepath):
        frames, self.loop = load_frames(filepath)
        self.data = np.full(frames.shape, np.nan)
        self.dt = frames[0][0]
        self.durations = frames[:, 0].tolist()
        cols = _raw_column_layout()
        st = {"root_pos": R.align_position(frames[:, 1:4]),
              "root_rot": R.align_rotation(frames[:, 4:8])}
        for j in BODY_JOINTS_IN_DP_ORDER:
            a, b = cols[j]
            st[j] = frames[:, a:b].copy() if DOF_DEF[j] == 1 else R.align_rotation(frames[:, a:b])
        self._states = st
        self.all_states = [{k: v[f] for k, v in st.items()} for f in range(frames.shape[0])]

    # -- stage 2: states -> data / data_config / data_vel (src/mujoco/mocap_v2.py:78-149) -----------
    def convert_raw_data(self):
        st = self._states
        F = st["root_pos"].shape[0]
        dur = np.asarray(self.durations, dtype=np.float64)
        dura = np.concatenate([dur[:1], dur[:-1]])          # frame k>0 uses durations[k-1]
        data = self.data
        cfg = np.zeros((F, 35))
        vel = np.zeros((F, 34))
        data[:, 0] = dura
        data[:, 1:4] = st["root_pos"]
        data[:, 4:8] = st["root_rot"]
        cfg[:, 0:3] = st["root_pos"]
        cfg[:, 3:7] = st["root_rot"]
        if F > 1:
            d1 = dura[1:]
            vel[1:, 0:3] = (data[1:, 1:4] - data[:-1, 1:4]) * 1.0 / d1[:, None]
            # NB argument order (current, previous): the reference's sign quirk is preserved
            vel[1:, 3:6] = R.rot_vel(data[1:, 4:8], data[:-1, 4:8], d1)
        od, oc, ov = 8, 7, 6
        for j in BODY_JOINTS:
            if DOF_DEF[j] == 1:
                data[:, od] = st[j][:, 0]
                cfg[:, oc] = st[j][:, 0]
                if F > 1:
                    vel[1:, ov] = (data[1:, od] - data[:-1, od]) * 1.0 / dura[1:]
                od, oc, ov = od + 1, oc + 1, ov + 1
            else:
                q = st[j]
                data[:, od:od + 4] = q
                if F > 1:
                    vel[1:, ov:ov + 3] = R.rot_vel(data[1:, od:od + 4], data[:-1, od:od + 4], dura[1:])
                cfg[:, oc:oc + 3] = R.euler_rxyz_from_quat_xyzw(q[:, [1, 2, 3, 0]])
                od, oc, ov = od + 4, oc + 3, ov + 3
        assert (od, oc, ov) == (44, 35, 34)
        # the reference keeps lists of per-frame arrays; 2-D arrays index identically (cfg[k], cfg[k][7:])
        self.data_config = cfg
        self.data_vel = vel

    def play(self, mocap_filepath):
        raise NotImplementedError("rendering is outside the accelerated path (SURVEY.md section 2, row 24)")
