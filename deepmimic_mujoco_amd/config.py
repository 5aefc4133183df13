"""Clip / model selection, mirroring `Config` (src/config.py:3-18).

The reference builds cwd-relative file paths and is edited in source; here `motion` may be changed at run time
(`Config.set_motion('walk')`) and the paths resolve to the bundled clip pack unless a real file exists.
`all_motions` is spelled out correctly (the reference's list is missing a comma, src/config.py:4-5)."""
import os


class Config(object):
    all_motions = ['backflip', 'cartwheel', 'crawl', 'dance_a', 'dance_b', 'getup_facedown', 'getup_faceup',
                   'jump', 'kick', 'punch', 'roll', 'run', 'spin', 'spinkick', 'walk']
    curr_path = os.getcwd()
    motion = 'dance_b'          # the committed value (src/config.py:9)
    env_name = "dp_env_v3"
    motion_folder = '/mujoco/motions'
    xml_folder = '/mujoco/humanoid_deepmimic/envs/asset'
    mocap_path = "%s%s/humanoid3d_%s.txt" % (curr_path, motion_folder, motion)
    xml_path = "%s%s/%s.xml" % (curr_path, xml_folder, env_name)

    @classmethod
    def set_motion(cls, motion):
        if motion not in cls.all_motions:
            raise ValueError("unknown motion %r" % motion)
        cls.motion = motion
        cls.mocap_path = "%s%s/humanoid3d_%s.txt" % (cls.curr_path, cls.motion_folder, motion)
