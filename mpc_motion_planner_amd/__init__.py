"""mpc_motion_planner_amd — host-side binding of libmpcmp.so (HIP / gfx950).

The product is the C-ABI shared library declared in include/mpcmp.h; this module is the thin ctypes
binding used by bench.py and the parity tests, plus `BatchMotionPlanner`, a batched mirror of the
reference's `MotionPlanner` façade (mpc_solver/motionPlanner.hpp:16-176).  There is no CPU fallback:
loading fails loudly when the HIP library is missing, and every solver call fails when no GPU is present.
(The scenario helpers of the robot wrapper — Jacobian, velocity maps, inverse kinematics — are host code in the
reference and here; they are not part of the batched solve.)
"""
from .capi import (STATUS_NAN, STATUS_NOT_PD, STATUS_XCH_DEAD, STATUS_QP_CAPPED, STATUS_OUTSIDE_TOL, STATUS_T_OUT_OF_BOX, STATUS_ARRIVED, STATUS_HARD,
                   Config, Info, Model, INFO_DTYPE, MpcmpError, DUAL_BASES, arm_models, build_library, default_config, default_limits,
                   default_model, forward_velocities, inverse_kinematic, inverse_velocities, lib, library_path,
                   model_from_urdf, models_from_urdf, num_nodes, time_nodes, tool_jacobian)
from .planner import BatchMotionPlanner, Solver

__all__ = ["STATUS_NAN", "STATUS_NOT_PD", "STATUS_XCH_DEAD", "STATUS_QP_CAPPED", "STATUS_OUTSIDE_TOL", "STATUS_T_OUT_OF_BOX", "STATUS_ARRIVED", "STATUS_HARD",
           "Config", "Info", "Model", "INFO_DTYPE", "MpcmpError", "build_library", "default_config",
           "default_limits", "default_model", "lib", "library_path", "model_from_urdf", "models_from_urdf", "num_nodes",
           "time_nodes", "BatchMotionPlanner", "Solver", "forward_velocities", "inverse_kinematic",
           "inverse_velocities", "tool_jacobian", "arm_models", "DUAL_BASES"]
