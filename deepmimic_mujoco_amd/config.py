"""RobotConfig / MotionConfig — host mirror of the reference's src/config.py:3-49.

Same attribute names; asset paths resolve inside this package instead of the
hard-coded ``~/Code/DeepMimic_mujoco/src`` (src/config.py:26,38).
"""
import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class RobotConfig:
    def __init__(self, robot="humanoid3d"):
        self.robot = robot
        if self.robot == "humanoid3d":  # src/config.py:6-13
            self.torso_body_name = "chest"
            self.lfoot_geom_name = "left_ankle"
            self.rfoot_geom_name = "right_ankle"
            self.floor_geom_name = "floor"
            self.extra_contact_geom_names = None
            self.endeffector_geom_names = ["left_ankle", "right_ankle", "left_wrist", "right_wrist"]
            self.low_z = 0.7
        elif self.robot == "unitree_g1":  # src/config.py:14-22 — names only: the G1 physics is not built;
            # the retargeting tool (retarget.py) needs the asset path, the envs refuse this robot (model.load_model)
            self.torso_body_name = "pelvis"
            self.lfoot_geom_name = "left_foot"
            self.rfoot_geom_name = "right_foot"
            self.floor_geom_name = "floor"
            self.extra_contact_geom_names = ["left_foot_lheel", "left_foot_rheel", "left_foot_ltoe", "left_foot_rtoe",
                                             "right_foot_lheel", "right_foot_rheel", "right_foot_ltoe", "right_foot_rtoe"]
            self.endeffector_geom_names = ["left_foot", "right_foot", "left_hand", "right_hand"]
            self.low_z = 0.4
        else:
            raise Exception("Unknown robot: %s" % (self.robot))
        self.env_name = "deepmimic_" + self.robot
        self.curr_path = _ASSETS
        self.xml_folder = ""
        self.xml_path = os.path.join(_ASSETS, "%s.xml" % self.env_name)


class MotionConfig(object):
    def __init__(self, motion=None, robot="humanoid3d"):
        # src/config.py:33-37 (the missing comma after 'getup_facedown' is the reference's)
        self.all_motions = ['backflip', 'cartwheel', 'crawl', 'dance_a', 'dance_b', 'getup_facedown'
                            'getup_faceup', 'jump', 'kick', 'punch', 'roll', 'run', 'spin', 'spinkick',
                            'walk']
        self.acyclical_motions = ["getup_faceup", "getup_facedown", "getup_facedown_slow",
                                  "getup_facedown_slow_FSI", "getup_facedown_towalk"]
        self.floor_motions = ["getup_faceup", "getup_facedown", "getup_facedown_slow",
                              "getup_facedown_slow_FSI", "getup_facedown_towalk"]
        self.curr_path = _ASSETS
        self.motion = 'walk' if motion is None else motion
        self.robot = robot
        self.env_name = "deepmimic_" + self.robot
        self.motion_folder = "motions"
        self.mocap_path = os.path.join(_ASSETS, "motions", "%s_%s.txt" % (self.robot, self.motion))
        self.xml_path = os.path.join(_ASSETS, "%s.xml" % self.env_name)
