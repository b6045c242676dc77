"""MI355X-native grasp-scoring engine for haf_grasping's sliding-window hot path.

Host interface mirrors the reference's CalcGraspPointsServer action (GraspInput -> GraspOutput); the compute
lives in libhafgrasp.so (hand-written gfx950 HIP kernels behind the C-ABI of include/hafgrasp.h)."""
from .capi import Engine, HafError, default_config, default_input, load_pcd  # noqa: F401
from .server import CalcGraspPointsServer, GraspInputMsg, GraspOutputMsg  # noqa: F401

__all__ = ["Engine", "HafError", "default_config", "default_input", "load_pcd", "CalcGraspPointsServer",
           "GraspInputMsg", "GraspOutputMsg"]
