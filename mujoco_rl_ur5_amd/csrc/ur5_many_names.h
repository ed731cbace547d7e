// ur5_many_names.h -- included first by ur5sim_many.hip: the many-object engine is the same source as the small-scene one
// compiled with other limits (-DUR5_MANY, ur5_devmodel.h), so every symbol with external linkage gets its own name here.
// libur5sim.so exports the public ur5_* names from ur5sim.hip only; they forward many-object handles to these.
#pragma once
#define ur5_sim ur5m_sim
#define ur5_last_error ur5m_last_error
#define ur5_create ur5m_create
#define ur5_destroy ur5m_destroy
#define ur5_num_envs ur5m_num_envs
#define ur5_nq ur5m_nq
#define ur5_nv ur5m_nv
#define ur5_nu ur5m_nu
#define ur5_reset ur5m_reset
#define ur5_reset_dev ur5m_reset_dev
#define ur5_reset_kernel ur5m_reset_kernel
#define ur5_kernel_ms_total ur5m_kernel_ms_total
#define ur5_set_state ur5m_set_state
#define ur5_get_state ur5m_get_state
#define ur5_set_ctrl ur5m_set_ctrl
#define ur5_get_ctrl ur5m_get_ctrl
#define ur5_step ur5m_step
#define ur5_move_group ur5m_move_group
#define ur5_stay ur5m_stay
#define ur5_move_ee ur5m_move_ee
#define ur5_ik ur5m_ik
#define ur5_grasp_attempt ur5m_grasp_attempt
#define ur5_grasp_attempt_dev ur5m_grasp_attempt_dev
#define ur5_grasp_attempt_reset_dev ur5m_grasp_attempt_reset_dev
#define ur5_grasp_rounds_dev ur5m_grasp_rounds_dev
#define ur5_aim_rule ur5m_aim_rule
#define ur5_sync ur5m_sync
#define ur5_set_order_dev ur5m_set_order_dev
#define ur5_set_stream ur5m_set_stream
#define ur5_last_launch_ms ur5m_last_launch_ms
#define ur5_get_counters ur5m_get_counters
#define ur5_body_xpos ur5m_body_xpos
#define ur5_render ur5m_render
#define ur5_render_dev ur5m_render_dev
#define ur5_state_device_ptr ur5m_state_device_ptr
#define ur5_forward_debug ur5m_forward_debug
#define ur5_profile_read ur5m_profile_read
#define ur5_run_kernel ur5m_run_kernel
#define ur5_render_kernel ur5m_render_kernel
#define ur5_render_pose_kernel ur5m_render_pose_kernel
#define Ur5GeomPose Ur5mGeomPose
#define ur5_cmodel ur5m_cmodel
#define ur5_smem ur5m_smem
#define Ur5DevModel Ur5mDevModel
