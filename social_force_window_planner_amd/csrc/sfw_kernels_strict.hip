// sfw_kernels_strict.hip — the K2 kernels once more, with the polynomial degrees of round 2 (asin 8 / exp 9), for
// SFW_PRECISION_F64_STRICT.  Since round 5 the default build evaluates the exponential at degree 9 too (asin 7 / exp 9), so
// this build differs by one degree of the angle's polynomial only.  Same source, compiled a second time: the degrees are
// compile-time constants of the Horner chains (an issue slot each), so the choice between the two is a choice between two
// sets of kernels, made per launch by sfw_capi.hip.  Every external symbol of sfw_kernels.hip is renamed for this
// translation unit; only sfw_launch_social_strict and sfw_launch_cycle_strict are used (the pose rollout, the footprint check and the selection do not
// evaluate a polynomial and exist once).
#define SFW_STRICT_BUILD 1
#define SFW_ASIN_DEG 8
#define SFW_EXP_DEG 9
#define sfw_launch_social sfw_launch_social_strict
#define sfw_launch_cycle sfw_launch_cycle_strict
#define sfw_cycle_applies sfw_strict_unused_cycle_applies
#define sfw_samples_per_wave sfw_strict_unused_samples_per_wave
#define sfw_social_organisation sfw_strict_unused_social_organisation
#define sfw_derive sfw_strict_unused_derive
#define sfw_social_flat_items sfw_strict_unused_social_flat_items
#define sfw_social_lds_bytes sfw_strict_unused_social_lds_bytes
#define sfw_pair_table_entries sfw_strict_unused_pair_table_entries
#define sfw_launch_pair_table sfw_strict_unused_launch_pair_table
#define sfw_rollout_is_fused sfw_strict_unused_rollout_is_fused
#define sfw_launch_rollout_poses sfw_strict_unused_launch_rollout_poses
#define sfw_launch_rollout_costmap sfw_strict_unused_launch_rollout_costmap
#define sfw_launch_rollout sfw_strict_unused_launch_rollout
#define sfw_launch_key_table sfw_strict_unused_launch_key_table
#define sfw_argmin_partials sfw_strict_unused_argmin_partials
#define sfw_launch_argmin sfw_strict_unused_launch_argmin
#include "sfw_kernels.hip"
