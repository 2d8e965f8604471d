"""MI355X-native DWA rollout + social-force scorer (one hot path of
robotics-upo/social_force_window_planner behind a C ABI)."""
from ._abi import (  # noqa: F401
    SFW_COST_INVALID,
    SFW_COST_SKIPPED,
    SFW_PRECISION_F32,
    SFW_PRECISION_F64,
    SfwAgent,
    SfwBest,
    SfwBestKey,
    SfwGoalArgs,
    SfwParams,
    SfwRobotState,
    default_params,
)
