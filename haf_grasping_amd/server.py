"""Host-side mirror of the reference's action interface for the hot path.

Reference: action/CalcGraspPointsServer.action:1-8 (goal GraspInput, result GraspOutput),
msg/GraspInput.msg:3-15, msg/GraspOutput.msg:1-7, and the server object CCalc_Grasppoints
(src/calc_grasppoints_action_server.cpp:107-229).  Field names and meanings are the reference's; the point
cloud is a float32 [N, 3] array already in the base frame (the server transforms it at :316 before the hot path).
"""
import dataclasses
from typing import Sequence

import numpy as np

from . import capi


@dataclasses.dataclass
class GraspInputMsg:
    """msg/GraspInput.msg.  grasp_area_length_* are declared 'in m' but the server uses them as integer
    centimetres including the 14 cm border (server.cpp:266-267; client.cpp:183-184 sends size + 14)."""
    input_pc: np.ndarray = None
    goal_frame_id: str = ""
    grasp_area_center: Sequence[float] = (0.0, 0.0, 0.0)
    grasp_area_length_x: float = 32.0
    grasp_area_length_y: float = 44.0
    max_calculation_time: float = 50.0
    show_only_best_grasp: bool = False
    threshold_grasp_evaluation: int = 0
    approach_vector: Sequence[float] = (0.0, 0.0, 1.0)
    gripper_opening_width: int = 1

    def to_c(self):
        return capi.default_input(grasp_area_center=tuple(self.grasp_area_center),
                                  grasp_area_length_x=float(self.grasp_area_length_x),
                                  grasp_area_length_y=float(self.grasp_area_length_y),
                                  approach_vector=tuple(self.approach_vector),
                                  max_calculation_time=float(self.max_calculation_time),
                                  show_only_best_grasp=int(bool(self.show_only_best_grasp)),
                                  threshold_grasp_evaluation=int(self.threshold_grasp_evaluation),
                                  gripper_opening_width=int(self.gripper_opening_width))


@dataclasses.dataclass
class GraspOutputMsg:
    """msg/GraspOutput.msg (header.frame_id = base frame, server.cpp:1386-1387)."""
    frame_id: str
    eval: int
    graspPoint1: tuple
    graspPoint2: tuple
    averagedGraspPoint: tuple
    approachVector: tuple
    roll: float

    def hypothesis_string(self, roll_step_deg=15):
        """The string the server publishes on /haf_grasping/grasp_hypothesis_with_eval (server.cpp:1384)."""
        g1, g2, av, avg = self.graspPoint1, self.graspPoint2, self.approachVector, self.averagedGraspPoint
        vals = [self.eval, *g1, *g2, *av, *avg]
        return " ".join("%g" % v for v in vals) + " %d" % int(round(np.degrees(self.roll) / roll_step_deg) * roll_step_deg)


class CalcGraspPointsServer:
    """Stands where CCalc_Grasppoints stands: construct once with the three data files (the ROS params of
    server.cpp:217-225), then execute(goal) per GraspInput.  No ROS here: the catkin shim that forwards the real
    action to the C-ABI is shown in INTEGRATION.md."""

    def __init__(self, feature_file_path, range_file_path, svmmodel_file_path, nr_features_without_shaf=302,
                 svm_with_probability=False, **cfg):
        if svm_with_probability:                     # the literal `false` of server.cpp:383, as a parameter (HAF_FLAG_PROBABILITY)
            cfg["flags"] = cfg.get("flags", 0) | capi.FLAG_PROBABILITY
        self.engine = capi.Engine(feature_file_path, range_file_path, svmmodel_file_path,
                                  nr_features_without_shaf=nr_features_without_shaf, **cfg)
        self.base_frame_id = "/base_link"            # server.cpp:294-301

    def execute(self, goal: GraspInputMsg) -> GraspOutputMsg:
        if goal.goal_frame_id:
            self.base_frame_id = goal.goal_frame_id
        out = self.engine.score(np.asarray(goal.input_pc, dtype=np.float32), goal.to_c())
        return GraspOutputMsg(self.base_frame_id, out["eval"], out["grasp_point1"], out["grasp_point2"],
                              out["averaged_grasp_point"], out["approach_vector"], out["roll"])

    def close(self):
        self.engine.close()
