/*
 * wbc_sim.h -- C-ABI of libwbc_amd.so: the MI355X-native replacement for the part of
 * the widowGo1 hot path that the reference delegates to Isaac Gym (closed source) and
 * to ~150 eager PyTorch ops per policy step.
 *
 * Every entry point names the reference interface it stands in for (paths relative to
 * the reference root; WG = legged_gym/legged_gym/envs/widowGo1/widowGo1.py,
 * BT = legged_gym/legged_gym/envs/base/base_task.py,
 * LR = legged_gym/legged_gym/envs/base/legged_robot.py,
 * RS = rsl_rl/rsl_rl/storage/rollout_storage.py).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative error code (text via wbc_last_error()); no exceptions cross the boundary; device
 * pointers are HIP device pointers on the device given at creation; `stream` is a
 * hipStream_t passed as void* (NULL = the legacy default stream) and all work is
 * asynchronous on that stream; the library creates no hidden streams or threads.
 * Quaternions are xyzw, float tensors are fp32, reset/episode-length buffers are int64,
 * as in the reference (BT:71-80).
 */
#ifndef WBC_SIM_H
#define WBC_SIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fixed sizes of the widowGo1 articulation ---- */
#define WBC_NB 19        /* moving bodies: floating root + 18 revolute joints */
#define WBC_NJ 18        /* revolute joints */
#define WBC_NDOF 20      /* simulator DoFs: 18 revolute + 2 locked prismatic fingers */
#define WBC_NACT 18      /* num_actions = num_torques (widowGo1_config.py:118-119) */
#define WBC_NRB 27       /* robot rigid bodies as the importer lists them */
#define WBC_NRB_ENV 28   /* + the free box actor (WG:384,542,546) */
#define WBC_NFEET 4
#define WBC_NCP 64       /* contact slots per env (one wavefront lane each): robot spheres vs terrain, box corners vs terrain,
                            the static pairs (arm spheres vs the trunk box, robot spheres vs the free box) and the DYNAMIC slots
                            that the self-collision broad phase promotes its hits into */
#define WBC_NSPH 28      /* the robot's contact spheres (their centres in frame F are cached once per substep) */
#define WBC_NLIMB 11     /* capsules of the self-collision broad phase: 4 thighs, 4 calves, upper arm, forearm, hand */
#define WBC_LIMB_RSUM_MAX 0.060f   /* an upper bound of (largest radius of limb a) + (largest radius of limb b) over all candidate pairs (second
                                      stage of the broad phase: the distance between the two shafts' segments against this + rest + margin) */
#define WBC_BOX_BODY WBC_NB   /* pseudo body index of the free box actor (WG:321-325,384) in cp_body / cp_body2 */
#define WBC_BOX_RB WBC_NRB    /* its row in the [N,28,...] rigid-body tensors (WG:544-548: the last one) */
#define WBC_NPROP 76     /* num_proprio (widowGo1_config.py:122) */
#define WBC_NPRIV 24     /* num_priv */
#define WBC_HIST 10      /* history_len */
#define WBC_NOBS 860     /* num_observations */
#define WBC_ADELAY_LEN 4 /* action_delay + 2 (WG:540) */
#define WBC_NREW 37      /* reward terms implemented (enum wbc_reward_term) */
#define WBC_NMETRIC 10   /* episode_metric_sums (WG:165) */
#define WBC_MAX_DEPTH 6

/* Articulated-body model, produced on the host from the URDF (replaces gym.load_asset +
 * get_asset_* getters, WG:285-294). All joint frames are unrotated (URDF rpy = 0). */
typedef struct {
  int32_t parent[WBC_NB];       /* parent moving body, -1 for the root */
  int32_t axis[WBC_NB];         /* 0/1/2 = x/y/z, -1 root */
  int32_t dof[WBC_NB];          /* simulator DoF index of joint i, -1 root */
  float joint_xyz[WBC_NB][3];   /* joint origin in the parent body frame */
  float mass[WBC_NB];
  float com[WBC_NB][3];
  float inertia[WBC_NB][6];     /* xx,yy,zz,xy,xz,yz about the com, body axes */
  float q_lower[WBC_NDOF], q_upper[WBC_NDOF];   /* URDF limits; lower==upper==0: unlimited */
  float qd_limit[WBC_NDOF];     /* URDF velocity limit */
  float effort[WBC_NDOF];       /* URDF effort limit -> torque_limits (LR:294-299) */
  int32_t rb_body[WBC_NRB];     /* moving body each importer rigid body rides on */
  float rb_offset[WBC_NRB][3];
  int32_t feet_rb[WBC_NFEET];   /* rigid-body indices of the feet, importer order FL,FR,RL,RR */
  int32_t gripper_rb;           /* wx250s/ee_gripper_link (WG:318) */
  /* Collision set (the URDF's <collision> blocks as sphere-swept primitives): ncp contacts in use.
   * Every contact has a sphere (centre cp_pos in the frame of moving body cp_body, radius cp_radius, riding on importer
   * rigid body cp_rb) and a partner selected by cp_kind:
   *   WBC_CP_TERRAIN  the terrain (plane or height grid);
   *   WBC_CP_BOX      a box fixed to moving body cp_body2 (rigid body cp_rb2): centre cp_a, half extents cp_b, its frame;
   * Pairs are two-body contacts (the robot's self-collision, asset.self_collisions = 0 = enabled, widowGo1_config.py:180; the robot
   * against the free box): the impulse acts on both bodies with opposite signs. Terrain contacts come first (the force sensors read contacts 0..3).
   * The free box actor (WG:321-325: a cube, density 1000) takes part under the pseudo body index WBC_BOX_BODY / rigid-body row
   * WBC_BOX_RB: its eight corner spheres against the terrain (cp_body = WBC_BOX_BODY, cp_pos in the box frame) and robot spheres
   * against it (kind WBC_CP_BOX with cp_body2 = WBC_BOX_BODY, cp_a = 0, cp_b = box_half).
   * Slots 0 .. ncp-1 are in use except those marked WBC_CP_NONE. The step kernel's layout rules (checked at wbc_sim_create): every
   * contact that involves the free box sits in slots 32..47 (one 16-lane row: their wrenches on the box are summed by a row
   * reduction) and nothing else does; keep what a walking robot normally touches with (feet, knees, trunk, arm) below 32 -- the
   * per-body loops walk the slots below 32 and those from 48 separately.
   *
   * Self-collision as configured (asset.self_collisions = 0: every pair of non-adjacent links collides): the pairs that can touch
   * inside the URDF's joint limits (tools/self_collision_reach.py: 44 limb pairs and 12 robot spheres against the free box; the legs
   * never reach the trunk box, the front and rear thighs never meet) are too many for one lane each, and almost never active. They are CANDIDATES: every lane carries one
   * pair descriptor (pr_*) -- its own pair for a static pair slot, otherwise a candidate -- and tests it against bounding spheres
   * in the same instructions (the broad phase; a limb pair that passes -- two legs standing side by side always do -- is then tested
   * segment against segment with the generous radius WBC_LIMB_RSUM_MAX before it counts as a hit). A candidate that passes is promoted into a free DYNAMIC slot (kind
   * WBC_CP_DYNAMIC: robot-vs-robot hits into the dynamic slots outside 32..47, in ascending order of candidate and slot;
   * robot-vs-free-box hits into the dynamic slots of the box row), where the exact test runs and, if the gap is inside the contact
   * margin, the contact is solved like any other pair. Hits beyond the free slots (17 outside the box row + 3 inside it) are dropped
   * and COUNTED (tensor WBC_T_DROPPED_HITS: per env, accumulated over the substeps; the tests assert it stays 0).
   * Primitives of the candidates, all in frame F from the cached sphere centres (cp_sph: the compact index of a robot sphere):
   *   limbs -- capsules between two sphere centres (thigh: hip-side end to knee, r 0.017; calf: knee to foot, r 0.008, with its end
   *   spheres knee r 0.02 / foot r 0.02; upper arm: shoulder joint to elbow, forearm: elbow to wrist, r 0.025; hand: wrist to gripper
   *   tip, r 0.02; equal end indices would make a single sphere); kind WBC_PR_LIMBS tests limb pr_a against
   *   limb pr_b as the union of shaft and end spheres (deepest feature pair wins: one contact per limb pair);
   *   kind WBC_PR_SPHERE_BOX tests robot sphere pr_a against the free box (knees, shins, trunk corners, wrist, elbow). */
  int32_t ncp;
  int32_t cp_body[WBC_NCP];
  float cp_pos[WBC_NCP][3];
  float cp_radius[WBC_NCP];
  int32_t cp_rb[WBC_NCP];       /* rigid body whose net_contact_force row receives the force */
  int32_t cp_kind[WBC_NCP];
  int32_t cp_body2[WBC_NCP];    /* -1 for terrain contacts */
  int32_t cp_rb2[WBC_NCP];      /* rigid body that receives the opposite force, -1 for terrain contacts */
  float cp_a[WBC_NCP][3], cp_b[WBC_NCP][3];
  float cp_radius2[WBC_NCP];
  int32_t cp_sph[WBC_NCP];      /* compact index (0 .. WBC_NSPH-1) of the robot sphere of a terrain slot / of a static pair's sphere; -1 otherwise */
  /* pair descriptor of every lane (broad phase): kind, operands, bounding reach (sum of the two bounding radii + contact margin) */
  int32_t pr_kind[WBC_NCP];     /* enum wbc_pair_kind */
  int32_t pr_a[WBC_NCP], pr_b[WBC_NCP];   /* WBC_PR_LIMBS: limb ids; WBC_PR_SPHERE_BOX / WBC_PR_STATIC: pr_a = robot sphere (compact index) */
  float pr_reach[WBC_NCP];
  int32_t nlimb;
  int32_t limb_s0[WBC_NLIMB], limb_s1[WBC_NLIMB];   /* compact sphere indices of the segment's ends (equal: a sphere) */
  float limb_radius[WBC_NLIMB], limb_cap0[WBC_NLIMB], limb_cap1[WBC_NLIMB];   /* shaft radius; end-sphere radii (0: none) */
  int32_t limb_body[WBC_NLIMB];                     /* moving body */
  int32_t limb_rb[WBC_NLIMB], limb_rb0[WBC_NLIMB], limb_rb1[WBC_NLIMB];   /* rigid body reported for a touch on the shaft / end sphere 0 / 1 */
  float pair_rest_offset;                           /* sim.physx.rest_offset for limb pairs (sphere radii already carry it) */
  /* pieces for per-env mass randomisation (WG:431-456) */
  float base_piece_mass, base_piece_com[3], base_piece_inertia[6];
  float base_rest_mass, base_rest_com[3], base_rest_inertia[6];
  int32_t gripper_body;
  float grip_piece_mass, grip_piece_com[3], grip_piece_inertia[6];
  float grip_rest_mass, grip_rest_com[3], grip_rest_inertia[6];
  /* the free box actor: half edge of the cube (box.box_size / 2, widowGo1_config.py:186), nominal mass (density 1000 x size^3,
   * WG:322; the per-env total is WBC_T_BOX_MASS), friction of its material (Isaac Gym's shape default 1.0) */
  float box_half, box_mass, box_friction;
  /* Sleeping (as PhysX puts resting actors to sleep: sleep threshold + wake counter): while the box's speed and (spin x half edge)
   * are below box_sleep_speed, at least three of its corners are within the contact offset of the terrain and no robot sphere is
   * within the contact offset of it, a per-env timer (WBC_T_BOX_SLEEP_TIMER) runs, otherwise it is zero; once it has reached
   * box_sleep_time the box is frozen -- no contacts, no gravity, zero velocity -- until a robot sphere touches it or it loses
   * its support (reset_idx re-places it in the air). box_sleep_speed <= 0: never sleeps. */
  float box_sleep_speed, box_sleep_time;
} wbc_model;

enum wbc_contact_kind { WBC_CP_NONE = -1 /* unused slot */, WBC_CP_TERRAIN = 0, WBC_CP_BOX = 1,
                        WBC_CP_LIMBS = 2 /* run time only: a dynamic slot holding a promoted limb pair */,
                        WBC_CP_DYNAMIC = 3 /* a free slot of the dynamic pool */ };
enum wbc_pair_kind { WBC_PR_NONE = 0, WBC_PR_STATIC = 1 /* the lane's own static pair (kind WBC_CP_BOX) */, WBC_PR_LIMBS = 2, WBC_PR_SPHERE_BOX = 3 };

enum wbc_reward_term {   /* the _reward_* methods WG defines (WG:1352-1469), then the base class's (LR = envs/base/legged_robot.py:832-922)
                            that work in the widowGo1 task. Not offered: orientation (LR:841-843 reads self.projected_gravity, which
                            WidowGo1 never creates: AttributeError in the reference, profiles/r04_reference_switches.txt), arm_orientation
                            (no such method), feet_stumble (the config key names no method; the method is _reward_stumble) */
  WBC_REW_ENERGY_SQUARE = 0, WBC_REW_SURVIVE, WBC_REW_TRACKING_LIN_VEL_X_L1,
  WBC_REW_TRACKING_ANG_VEL_YAW_EXP, WBC_REW_HIP_ACTION_L2, WBC_REW_FOOT_CONTACTS_Z,
  WBC_REW_TRACKING_EE_SPHERE, WBC_REW_ARM_ENERGY_ABS_SUM,
  WBC_REW_TRACKING_EE_CART, WBC_REW_TRACKING_EE_ORN, WBC_REW_TRACKING_EE_ORN_RY,
  WBC_REW_LEG_ENERGY_ABS_SUM, WBC_REW_LEG_ENERGY_SUM_ABS, WBC_REW_LEG_ACTION_L2,
  WBC_REW_LEG_ENERGY, WBC_REW_TRACKING_LIN_VEL, WBC_REW_TRACKING_LIN_VEL_X_EXP,
  WBC_REW_TRACKING_ANG_VEL_YAW_L1, WBC_REW_TRACKING_LIN_VEL_Y_L2,
  WBC_REW_TRACKING_LIN_VEL_Z_L2, WBC_REW_TORQUES, WBC_REW_COLLISION,
  /* base class */
  WBC_REW_LIN_VEL_Z, WBC_REW_ANG_VEL_XY, WBC_REW_DOF_VEL, WBC_REW_DOF_ACC, WBC_REW_ACTION_RATE,
  WBC_REW_TERMINATION,   /* added AFTER the only_positive_rewards clip (WG:184-188, 200-203) */
  WBC_REW_DOF_POS_LIMITS, WBC_REW_DOF_VEL_LIMITS, WBC_REW_TORQUE_LIMITS, WBC_REW_TRACKING_ANG_VEL,
  WBC_REW_FEET_AIR_TIME, /* stateful: WBC_T_FEET_AIR_TIME / WBC_T_LAST_CONTACTS, advanced only while the term is active */
  WBC_REW_STUMBLE, WBC_REW_STAND_STILL, WBC_REW_FEET_CONTACT_FORCES,
  WBC_REW_BASE_HEIGHT    /* with terrain.measure_heights = False (measured_heights = 0, WG:639): (z - base_height_target)^2 */
};

enum wbc_metric {        /* WG:165 order */
  WBC_MET_LEG_ENERGY_ABS_SUM = 0, WBC_MET_TRACKING_LIN_VEL_X_L1, WBC_MET_TRACKING_ANG_VEL_YAW_EXP,
  WBC_MET_TRACKING_EE_CART, WBC_MET_TRACKING_EE_SPHERE, WBC_MET_TRACKING_EE_ORN,
  WBC_MET_LEG_ACTION_L2, WBC_MET_TORQUE, WBC_MET_ENERGY_SQUARE, WBC_MET_FOOT_CONTACTS_Z
};

/* Task constants: WidowGo1RoughCfg + LeggedRobotCfg.sim resolved to numbers. */
typedef struct {
  /* sim (legged_robot_config.py:182-199) */
  float sim_dt;                 /* 0.005 */
  int32_t decimation;           /* 4 */
  float gravity[3];
  /* contact / limit model of THIS framework's physics spec */
  float contact_margin;         /* physx.contact_offset 0.01 */
  float contact_erp;            /* fraction of penetration removed per step */
  float max_depenetration_vel;  /* physx.max_depenetration_velocity 1.0 */
  float terrain_friction;       /* terrain.static_friction 1.0 */
  float limit_kappa, limit_delta; /* joint-limit stop gains, in units of D/dt^2 and D/dt */
  int32_t contact_iters;
  float joint_armature[WBC_NACT]; /* added to each joint's articulated inertia D: dt*Kd + dt^2*Kp makes
                                     the task's PD law (WG:1281) linearly implicit */
  /* control (WG:1262-1295, widowGo1_config.py:163-173) */
  float clip_actions;
  float action_scale[WBC_NACT];
  float p_gains[WBC_NACT], d_gains[WBC_NACT];
  float default_dof_pos[WBC_NDOF];
  float torque_limits[WBC_NDOF];
  int32_t action_delay;         /* 2; -1 disables the FIFO */
  /* observations (WG:966-1001) */
  float obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, clip_obs;
  float commands_scale[3];
  /* episode, termination (WG:937-963) */
  int32_t max_episode_length;   /* 500 */
  float term_rp_threshold;      /* 0.2, hard-coded at WG:945-946 */
  float term_z_threshold;       /* 0.325 */
  uint32_t term_contact_rb_mask;     /* bit rb: rigid body rb is in asset.terminate_after_contacts_on (WG:305-306,940: |net force| > 1 N
                                        ends the episode); shipped config: empty */
  uint32_t penalize_contact_rb_mask; /* asset.penalize_contacts_on (WG:299-300; _reward_collision counts |net force| > 0.1 N, LR:865-867) */
  int32_t resample_interval;    /* int(3.0/0.02)=150 (WG:922) */
  int32_t push_interval;        /* 150 (WG:119); <=0 disables */
  float max_push_vel;
  float lin_vel_x_clip, ang_vel_yaw_clip;
  /* EE goal sampler (WG:1297-1350, widowGo1_config.py:46-85) */
  float goal_collision_lower[3], goal_collision_upper[3];
  float goal_underground_limit;
  int32_t goal_collision_samples;     /* 10 */
  float goal_delta_orn_range[3][2];
  float sphere_error_scale[3], orn_error_scale[3];
  float z_invariant_offset;           /* 0.53 (WG:597) */
  int32_t goal_command_cart;          /* goal_ee.command_mode == 'cart' (WG:589-593): curr_ee_goal is the Cartesian goal -- observation
                                         entries 70..72 (WG:980) and the sign tests of the roll / pitch termination (WG:945-946) read it
                                         instead of the spherical one; 0 = 'sphere' (the shipped config) */
  /* rewards */
  float tracking_sigma, tracking_ee_sigma;
  int32_t only_positive_rewards;
  /* the base class's terms: soft joint limits (LR:301-304: centre +- half range * rewards.soft_dof_pos_limit), velocity limits
   * * soft_dof_vel_limit (LR:882), torque limits * soft_torque_limit (LR:886), rewards.max_contact_force, base_height_target */
  float soft_dof_lower[WBC_NDOF], soft_dof_upper[WBC_NDOF], soft_dof_vel_limit[WBC_NDOF], soft_torque_limit[WBC_NDOF];
  float max_contact_force, base_height_target;
  /* resets (WG:757-828) */
  float base_init_state[13];
  float origin_perturb_range, init_vel_perturb_range;
  float dof_reset_lo, dof_reset_hi;   /* 0.8, 1.2 */
  float box_origin_x, box_origin_z;
  /* flat ground height unless a heightfield is attached */
  float ground_z;
} wbc_task_cfg;

/* Host scalars the reference recomputes in update_command_curriculum (WG:678-692). */
typedef struct {
  float lin_vel_x_range[2], ang_vel_yaw_range[2];
  float goal_l_range[2], goal_p_range[2], goal_y_range[2];
  float leg_reward_scale[WBC_NREW];   /* rewards.scales: CURRENT value of each term's scale (WG:176-178) */
  float arm_reward_scale[WBC_NREW];   /* rewards.arm_scales */
  /* bit t set = term t's _reward_ function is in the list built at construction from the config's non-zero scales
   * (_prepare_reward_function, WG:128-157): it is evaluated every step -- episode sum += term * current scale, metric side
   * effect applied -- even while a scheduled scale is 0 */
  uint64_t leg_active_mask, arm_active_mask;
} wbc_curriculum;

/* Device tensors owned by the sim; ids for wbc_sim_get_tensor. Shapes at N envs. */
enum wbc_tensor_id {
  WBC_T_ROOT_STATES = 0,   /* f32 [N,2,13]  acquire_actor_root_state_tensor  (WG:505,523) */
  WBC_T_DOF_STATE,         /* f32 [N,20,2]  acquire_dof_state_tensor         (WG:506,526) */
  WBC_T_NET_CONTACT_FORCE, /* f32 [N,28,3]  acquire_net_contact_force_tensor (WG:507,542) */
  WBC_T_RIGID_BODY_STATE,  /* f32 [N,28,13] acquire_rigid_body_state_tensor  (WG:508,546) */
  WBC_T_FORCE_SENSOR,      /* f32 [N,4,6]   acquire_force_sensor_tensor      (WG:511,522) */
  WBC_T_TORQUES,           /* f32 [N,20]    self.torques                     (WG:619,1176) */
  WBC_T_OBS_BUF,           /* f32 [N,860]   obs_buf                          (WG:992,1196) */
  WBC_T_OBS_HISTORY,       /* f32 [N,10,76] obs_history_buf                  (WG:539,994) */
  WBC_T_ACTION_HISTORY,    /* f32 [N,4,18]  action_history_buf               (WG:540,1167) */
  WBC_T_ACTIONS,           /* f32 [N,18]    self.actions (delayed, sim order)(WG:1173) */
  WBC_T_LAST_ACTIONS,      /* f32 [N,18]                                     (WG:623,908) */
  WBC_T_LAST_DOF_VEL,      /* f32 [N,20]                                     (WG:624,909) */
  WBC_T_LAST_ROOT_VEL,     /* f32 [N,6]                                      (WG:625,910) */
  WBC_T_COMMANDS,          /* f32 [N,3]                                      (WG:627) */
  WBC_T_GOAL_STATE,        /* f32 [N,24]  ee_start_sphere,ee_goal_sphere,ee_goal_cart,
                              curr_ee_goal_sphere,curr_ee_goal_cart,ee_goal_delta_orn_euler,
                              ee_goal_orn_euler (3 each), goal_timer, traj_timesteps,
                              traj_total_timesteps                           (WG:574-583) */
  WBC_T_REW_BUF,           /* f32 [N]                                        (BT:72) */
  WBC_T_ARM_REW_BUF,       /* f32 [N] */
  WBC_T_RESET_BUF,         /* i64 [N]                                        (BT:74) */
  WBC_T_TIME_OUT_BUF,      /* u8  [N]  (bool)                                (BT:76) */
  WBC_T_EPISODE_LENGTH,    /* i64 [N]                                        (BT:75) */
  WBC_T_EPISODE_SUMS,      /* f32 [N,37] per-term sums (WBC_NREW), zeroed on reset (WG:162) */
  WBC_T_METRIC_SUMS,       /* f32 [N,10]                                     (WG:166) */
  WBC_T_EPISODE_SUMS_DONE, /* f32 [N,37] sums at the moment of reset (for extras) (WG:743-746) */
  WBC_T_METRIC_SUMS_DONE,  /* f32 [N,10]                                     (WG:748-750) */
  WBC_T_BASE_LIN_VEL,      /* f32 [N,3]                                      (WG:880) */
  WBC_T_BASE_ANG_VEL,      /* f32 [N,3]                                      (WG:881) */
  WBC_T_MASS_PARAMS,       /* f32 [N,5]  mass_params_tensor                  (WG:354,455) */
  WBC_T_FRICTION,          /* f32 [N]    friction_coeffs_tensor              (WG:400) */
  WBC_T_MOTOR_STRENGTH,    /* f32 [N,18]                                     (WG:403) */
  WBC_T_ENV_ORIGINS,       /* f32 [N,3]                                      (WG:212) */
  WBC_T_BOX_DELTA_Y,       /* f32 [N]                                        (WG:226) */
  WBC_T_BODY_PARAMS,       /* f32 [N,20] per-env composite root + gripper (mass, com, inertia6) */
  WBC_T_RESET_TRAVEL,      /* f32 [N,2]  at the moment of an env's reset: ||root_xy - env_origin_xy|| and ||commands[:2]||,
                              the two quantities _update_terrain_curriculum reads before reset_idx overwrites them (LR:431-435) */
  WBC_T_BOX_MASS,          /* f32 [N]    total mass of the env's box actor: nominal + box.added_mass_range draw (WG:458-466) */
  WBC_T_BOX_SLEEP_TIMER,   /* f32 [N]    substeps the box actor has been at rest (asleep from box_sleep_time / sim_dt on) */
  WBC_T_FEET_AIR_TIME,     /* f32 [N,4]  feet_air_time (WG:633, LR:898-909), zeroed on reset (WG:734) */
  WBC_T_LAST_CONTACTS,     /* f32 [N,4]  last_contacts as 0 / 1 (WG:626, LR:902-903) */
  WBC_T_DROPPED_HITS,      /* f32 [N]    broad-phase hits that found no free dynamic contact slot, accumulated over all substeps since the sim was
                              created (diagnostic: a non-zero value means a self-collision / box contact was not simulated) */
  WBC_T_COUNT
};
enum wbc_dtype { WBC_F32 = 0, WBC_I64 = 1, WBC_U8 = 2 };

typedef struct wbc_sim wbc_sim;

const char* wbc_last_error(void);

/* ---- asset: what gym.load_asset + the get_asset_* getters (WG:268-294) and the config resolution of WidowGo1._parse_cfg /
 * _init_buffers (WG:78-121, 498-672) hand to the reference, as ONE binary file -- so that a binding in any language can create a
 * sim without this repository's Python: wbc_asset_load -> wbc_asset_model / _task_cfg / _curriculum -> (edit the documented fields
 * of its own copies: gains, thresholds, reward scales ...) -> wbc_sim_create. The packaged widowGo1 asset with the shipped config is
 * deep-whole-body-control_amd/wbc_amd/assets/widowgo1_default.wbcasset (written by tools/make_asset.py from the URDF tables and
 * WidowGo1RoughCfg; a test keeps it equal to what the Python host path builds). File layout: magic "WBCASSET1", the three struct
 * sizes (checked against this library's: -3 on a mismatch = another ABI version), dof / rigid-body counts, the structs (wbc_model,
 * wbc_task_cfg, wbc_curriculum before and after the first update_command_curriculum call), then the names (64-byte fields). */
typedef struct wbc_asset wbc_asset;
int wbc_asset_load(const char* path, wbc_asset** out);
void wbc_asset_free(wbc_asset* asset);
int wbc_asset_dof_count(const wbc_asset* asset);                       /* gym.get_asset_dof_count (WG:287) */
int wbc_asset_rigid_body_count(const wbc_asset* asset);                /* gym.get_asset_rigid_body_count (WG:288) */
const char* wbc_asset_dof_name(const wbc_asset* asset, int i);         /* gym.get_asset_dof_names (WG:293) */
const char* wbc_asset_rigid_body_name(const wbc_asset* asset, int i);  /* gym.get_asset_rigid_body_names (WG:292) */
/* gym.get_asset_dof_properties (WG:289, LR:279-305): lower / upper / velocity / effort, each dof_count floats (NULL: skipped) */
int wbc_asset_dof_properties(const wbc_asset* asset, float* lower, float* upper, float* velocity, float* effort);
const wbc_model* wbc_asset_model(const wbc_asset* asset);
const wbc_task_cfg* wbc_asset_task_cfg(const wbc_asset* asset);
/* which = 0: as the task object carries them before the first update_command_curriculum call; 1: after it (the shipped schedules
 * saturate on the first call, WG:678-692) */
const wbc_curriculum* wbc_asset_curriculum(const wbc_asset* asset, int which);

/* Bytes of device memory a sim of `num_envs` needs. */
size_t wbc_sim_arena_bytes(int num_envs);

/* gymapi.acquire_gym + create_sim + load_asset + the create_env/create_actor loop +
 * prepare_sim (BT:42,86-87; WG:234,285,355-392). `arena` is caller-provided device memory
 * of >= wbc_sim_arena_bytes(num_envs) bytes (e.g. a torch allocation, so that the tensors
 * below are zero-copy views of it); NULL lets the library hipMalloc its own. 1 <= num_envs <= 2^22 (the kernels index a tensor's
 * rows with 32-bit element offsets); anything else is -1 with a message in wbc_last_error(). */
int wbc_sim_create(const wbc_model* model, const wbc_task_cfg* cfg, int num_envs, int hip_device,
                   uint64_t seed, void* arena, size_t arena_bytes, wbc_sim** out);
int wbc_sim_destroy(wbc_sim* sim);

/* acquire_*_tensor + gymtorch.wrap_tensor (WG:505-551): device pointer, shape, dtype. */
int wbc_sim_get_tensor(wbc_sim* sim, int id, void** dev_ptr, int64_t shape[4], int* ndim, int* dtype);

/* Batched replacement for the O(N) per-env Python loop that sets shape friction and
 * randomised rigid-body properties (WG:365-389, 431-496) and motor strengths (WG:402-408).
 * Host pointers, N entries each (motor_strength N*18, base_dcom N*3, env_origins N*3); box_dmass = the box actor's added
 * mass (_box_process_rigid_body_props, WG:458-466), NULL = 0. */
int wbc_sim_set_env_params(wbc_sim* sim, const float* friction, const float* base_dmass,
                           const float* base_dcom, const float* gripper_dmass,
                           const float* motor_strength, const float* env_origins,
                           const float* box_delta_y, const float* traj_timesteps,
                           const float* traj_total_timesteps, const float* box_dmass);

/* gym.add_triangle_mesh for the regular-grid terrain (WG:242-252): the int16 height samples
 * the trimesh is built from, host pointer, rows*cols; NULL restores the flat plane. */
int wbc_sim_set_heightfield(wbc_sim* sim, const int16_t* heights, int rows, int cols,
                            float horizontal_scale, float vertical_scale,
                            float tx, float ty, float tz);

/* update_command_curriculum results (WG:678-692). */
int wbc_sim_set_curriculum(wbc_sim* sim, const wbc_curriculum* cur);

/* WidowGo1.step (WG:1156-1199) as ONE launch: action reorder/clip/delay FIFO, 4x
 * {_compute_torques, simulate}, post_physics_step (EE goal, commands, pushes, termination,
 * rewards, resets, observations), obs clipping. actions: device f32 [N,18], policy order. */
int wbc_sim_step(wbc_sim* sim, const float* actions_dev, void* stream);
/* The same step with the observation rows written to obs_out_dev (f32 [N,860], e.g. the rollout storage slot of the next
 * transition: what `self.observations[self.step].copy_(transition.observations)`, rollout_storage.py:66, would copy)
 * instead of WBC_T_OBS_BUF; obs_out_dev == NULL is wbc_sim_step. */
int wbc_sim_step_to(wbc_sim* sim, const float* actions_dev, float* obs_out_dev, void* stream);
/* The step with PPO.process_env_step's tensor work folded in (rsl_rl/algorithms/ppo.py:129-141, rollout_storage.py:70-72; what
 * wbc_rollout_store does as a separate launch): out_rewards[n] = (rew[n], arm_rew[n]) + gamma * values[n] * time_out[n],
 * out_dones[n] = reset_buf[n] != 0, written to the rollout storage's slots of this transition. values_dev: f32 [N,2], the
 * critic's output for the observation the actions were computed from. obs_out_dev as in wbc_sim_step_to; any of the two
 * groups may be NULL. */
int wbc_sim_step_rollout(wbc_sim* sim, const float* actions_dev, float* obs_out_dev, const float* values_dev, float gamma,
                         float* out_rewards_dev, uint8_t* out_dones_dev, void* stream);

/* BaseTask.reset() first half: reset_idx(all envs, start=True) (BT:127-131, WG:695-754). */
int wbc_sim_reset_all(wbc_sim* sim, void* stream);

/* gym.set_dof_actuation_force_tensor (WG:1183): device f32 [N,20]. */
int wbc_sim_set_dof_forces(wbc_sim* sim, const float* torques_dev, void* stream);
/* gym.simulate (WG:1184): one physics substep with the torques currently set. */
int wbc_sim_simulate(wbc_sim* sim, void* stream);
/* gym.set_actor_root_state_tensor / set_dof_state_tensor (WG:787,814,827): device pointers
 * with the layouts of WBC_T_ROOT_STATES / WBC_T_DOF_STATE; passing the sim's own tensor is a
 * no-op copy. The _indexed forms (LR:389-391,410-412) take int32 env ids on the device. */
int wbc_sim_set_root_state(wbc_sim* sim, const float* root_dev, void* stream);
int wbc_sim_set_dof_state(wbc_sim* sim, const float* dof_dev, void* stream);
int wbc_sim_set_root_state_indexed(wbc_sim* sim, const float* root_dev, const int32_t* env_ids_dev, int n, void* stream);
int wbc_sim_set_dof_state_indexed(wbc_sim* sim, const float* dof_dev, const int32_t* env_ids_dev, int n, void* stream);
/* gym.refresh_*_tensor (WG:513-519,870-873,1187-1191): state is resident, these return 0;
 * refresh_rigid_body_state recomputes forward kinematics after a state write. */
int wbc_sim_refresh_dof_state(wbc_sim* sim);
int wbc_sim_refresh_root_state(wbc_sim* sim);
int wbc_sim_refresh_net_contact_force(wbc_sim* sim);
int wbc_sim_refresh_force_sensor(wbc_sim* sim);
int wbc_sim_refresh_rigid_body_state(wbc_sim* sim, void* stream);

/* Step counter (common_step_counter, WG:613,876) get/set, for checkpoint/resume and tests. */
int wbc_sim_get_step_counter(wbc_sim* sim, int64_t* out);
int wbc_sim_set_step_counter(wbc_sim* sim, int64_t value);

/* RolloutStorage.compute_returns (RS:136-150) as one launch: reverse-time GAE over both
 * reward channels with shared dones, then advantages = returns - values. Device pointers:
 * rewards, values, returns, advantages f32 [T,N,2]; dones u8 [T,N]; last_values f32 [N,2].
 * stats_dev (f64 [wbc_gae_workspace_doubles(N)]; the first three entries receive count, sum and
 * sum of squares of the raw advantages) is filled for the joint normalisation (RS:150), which
 * wbc_gae_normalize applies with the (optionally all-reduced) statistics. */
int wbc_gae_compute(const float* rewards, const float* values, const uint8_t* dones,
                    const float* last_values, float* returns, float* advantages, double* stats_dev,
                    int T, int N, float gamma, float lam, void* stream);
/* (advantages - mean) / (std + 1e-8) over `total` = T*N*2 elements with the statistics in stats_dev[0..2]. */
int wbc_gae_normalize(float* advantages, const double* stats_dev, int64_t total, void* stream);
/* Number of doubles stats_dev must hold for N envs (3 statistics + per-block partials). */
int wbc_gae_workspace_doubles(int N);

/* Side jobs: a small per-step reduction that ANOTHER launch carries as a few extra workgroups instead of being a launch (and an
 * inter-kernel dependency stall) of its own. wbc_sim_episode_stats_job describes the work of wbc_sim_episode_stats_track without
 * launching it; wbc_policy_act_job (the policy inference that follows every env step in a rollout anyway) executes it next to
 * its own workgroups, wbc_side_job_run executes it stand-alone. The job holds device pointers into the sim's tensors: it must be
 * executed before the sim's next step. */
typedef struct {
  const float* ep_done; const float* met_done; const int64_t* reset_buf; const float* prev; const float* rew; const float* arm_rew;
  float* out; float* track_state;
  int32_t n, track_cap, nblocks;     /* nblocks: workgroups the job needs (one per statistics column + 1 with a tracker state) */
  float scale;
} wbc_side_job;
int wbc_sim_episode_stats_job(wbc_sim* sim, float scale, const float* prev, float* out, float* track_state, int track_cap,
                              wbc_side_job* job);
int wbc_side_job_run(const wbc_side_job* job, void* stream);

/* The policy side of one rollout step, PPO.act (rsl_rl/algorithms/ppo.py:115-127) with the privileged
 * latent: Actor.forward (actor_critic.py:204-221), Critic.forward (:281-286), the action sample
 * mean + std * eps (Normal.sample, :337-339) and get_actions_log_prob (:341-345) in ONE launch on fp32
 * MFMA. `params`: 33 device pointers in state_dict order of the layers used (struct PolicyParams in
 * csrc/wbc_mlp.h: priv_encoder.{0,2}, actor_backbone.0, leg head {0,2,4}, arm head {0,2,4},
 * critic_backbone.0, critic leg head {0,2,4}, critic arm head {0,2,4}, each weight then bias, then std).
 * `wpack`: wbc_policy_pack_floats() floats holding the weights in MFMA-fragment order; refresh it with
 * wbc_policy_pack whenever the parameters changed. obs f32 [rows,860]; eps f32 [rows,18] standard
 * normals (NULL: act on the mean); outputs actions/mean f32 [rows,18], logp/values f32 [rows,2].
 * `latent` f32 [rows,20] or NULL: if given (student path, hist_encoding=True, actor_critic.py:206-209) it replaces
 * the privileged encoder's output (e.g. the result of wbc_hist_latent). */
int wbc_policy_pack_floats(void);
int wbc_policy_pack(const void* const* params, float* wpack, void* stream);
int wbc_policy_act(const void* const* params, const float* wpack, const float* obs, const float* latent,
                   const float* eps, float* actions, float* mean, float* logp, float* values, int num_rows,
                   void* stream);
/* wbc_policy_act + a side job (may be NULL) executed by extra workgroups of the same launch. */
int wbc_policy_act_job(const void* const* params, const float* wpack, const float* obs, const float* latent,
                       const float* eps, float* actions, float* mean, float* logp, float* values, int num_rows,
                       const wbc_side_job* job, void* stream);

/* StateHistoryEncoder forward without gradient (rsl_rl/modules/actor_critic.py:39-84, tsteps = 10), the regulariser
 * target of PPO.update (ppo.py:174-176). params: 8 device pointers (encoder.0.weight [30,76], .bias,
 * conv_layers.0.weight [20,30,4], .bias, conv_layers.2.weight [10,20,2], .bias, linear_output.0.weight [20,30],
 * .bias); reads obs[:, 100:860] of f32 [rows,860]; writes out f32 [rows,20]. ELU activations. */
int wbc_hist_latent(const void* const* params, const float* obs, float* out, int rows, void* stream);

/* One PPO.update() minibatch (rsl_rl/algorithms/ppo.py:163-246, teacher path, no torque supervision): gathers
 * the rows `idx` of the flat [T*N, ...] rollout tensors, runs actor + critic forward, the clipped surrogate
 * with Advantage Mixing (:199-206), the (clipped) value loss (:209-216) and the ROA regulariser (:174-179),
 * and back-propagates; writes into `grad` (wbc_ppo_grad_floats() floats) the gradient of
 * surrogate + value_coef*value_loss + roa_coef*priv_reg in the order: for each of the 16 layers of
 * `params` (same table as wbc_policy_act) weight then bias; std[18]; then the three loss SUMS (surrogate and
 * value over B*2 entries, priv_reg over B). `hist_latent` f32 [T*N,20]: history-encoder latent of every
 * stored row (constant during update()). `workspace`: wbc_ppo_workspace_floats(B) floats. `loss_accum` (device, 3 floats,
 * or NULL): the three loss sums are also ADDED to it (an update's running totals without a launch of its own).
 * Also leaves, at workspace + wbc_ppo_sq_partials_offset(B), partial sums of squares of the gradient it wrote
 * (wbc_ppo_clip_adam's `sq_partials`). Deterministic. B < 338 000 rows (32-bit offsets into the workspace); -3 beyond. */
int wbc_ppo_minibatch_grad(const void* const* params, const float* obs, const float* actions,
                           const float* old_values, const float* advantages, const float* returns,
                           const float* old_logp, const float* hist_latent, const int64_t* idx, int B,
                           float clip, float value_coef, float mixing, float roa_coef,
                           int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream);
/* The same call without its weight-pack launch (5.9 us + a launch gap of every minibatch): valid while `workspace` still holds the
 * weight streams that a wbc_ppo_minibatch_grad call for the same B packed into it and every change of the parameters since then
 * was a wbc_ppo_clip_adam_packed(..., workspace, B) step (which keeps the streams current). The library keeps a record per workspace
 * (B, params[0]) of which streams are current: a call whose (workspace, B, params[0]) does not match it -- a different B, a
 * reallocated workspace, another parameter set, a plain wbc_ppo_clip_adam step in between -- packs afresh instead of computing
 * gradients against stale weights. What the library cannot see is a write to the parameters by other code (a checkpoint load, a
 * broadcast): follow it with wbc_ppo_minibatch_grad (or wbc_ppo_pack_invalidate(workspace); NULL drops every record). */
int wbc_ppo_minibatch_grad_packed(const void* const* params, const float* obs, const float* actions,
                                  const float* old_values, const float* advantages, const float* returns,
                                  const float* old_logp, const float* hist_latent, const int64_t* idx, int B,
                                  float clip, float value_coef, float mixing, float roa_coef,
                                  int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream);
int wbc_ppo_pack_invalidate(const float* workspace);
size_t wbc_ppo_sq_partials_offset(int B);
/* nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() (ppo.py:243-246) for the 33 parameters of
 * `params`, reading their gradients from grad[0 : wbc_ppo_grad_floats()-3] (scaled in place by the clip factor);
 * exp_avg / exp_avg_sq are flat Adam moments in the same layout. step_size = lr/(1-beta1^t), bc2_sqrt =
 * sqrt(1-beta2^t). max_norm <= 0 disables the clip. grad_scale (> 0) multiplies the gradient before the clip: 1 on one GPU,
 * 1 / world_size after the SUM all-reduce of the sharded learner (the mean over ranks without a separate launch).
 * sq_partials: the partial sums of squares wbc_ppo_minibatch_grad left for THIS gradient (pass them only while grad is
 * exactly what that call wrote -- not after an all-reduce), or NULL: the same partials are then recomputed from grad (same
 * blocks, same order: the two ways give the same bits for the same gradient).
 * workspace: >= wbc_ppo_clip_adam_workspace_floats() floats. Deterministic. */
int wbc_ppo_clip_adam_workspace_floats(void);
int wbc_ppo_clip_adam(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm,
                      float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                      const float* sq_partials, float* workspace, void* stream);
/* wbc_ppo_clip_adam, and every updated parameter is also written to its copies in the weight streams inside `mb_workspace` (the
 * wbc_ppo_minibatch_grad workspace for B rows): see wbc_ppo_minibatch_grad_packed. A workspace whose streams were not packed from
 * these parameters for this B is left alone (the step is then a plain wbc_ppo_clip_adam). -4: the scatter table could not be built. */
int wbc_ppo_clip_adam_packed(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm,
                             float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                             const float* sq_partials, float* workspace, float* mb_workspace, int B, void* stream);
/* One minibatch of PPO.update_dagger (rsl_rl/algorithms/ppo.py:265-291): the history encoder's forward
 * (rsl_rl/modules/actor_critic.py:39-84), loss = mean over rows of ||target - latent||_2, backward and weight gradients
 * for the `rows` rows obs[idx[r]] (obs f32 [batch, 860], target f32 [batch, 20] = the privileged latents, idx i64 [rows]).
 * params: the 8 tensors of wbc_hist_latent. grad (out): wbc_hist_train_grad_floats() floats = the flat gradient in
 * history_encoder.parameters() order (5760) followed by the minibatch's SUM of row norms (divide by rows for the loss).
 * workspace: wbc_hist_train_workspace_floats() floats. Deterministic (fixed-order reduction). */
int wbc_hist_train_grad(const void* const* params, const float* obs, const float* target, const long long* idx, int rows,
                        float* workspace, float* grad, void* stream);
int wbc_hist_train_grad_floats(void);
size_t wbc_hist_train_workspace_floats(void);
/* clip_grad_norm_ + Adam.step (ppo.py:283-284) of the history encoder on the flat buffers, as wbc_ppo_clip_adam.
 * grad_was_reduced != 0: grad was changed since wbc_hist_train_grad (all-reduce over ranks); the squares of the reduced
 * gradient are re-derived by a launch of their own BEFORE the one that rescales grad in place (every block of the Adam
 * launch sees the same norm). workspace: the one passed to wbc_hist_train_grad (its squared-gradient tail is rewritten). */
int wbc_hist_clip_adam(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                       float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, int grad_was_reduced,
                       float* workspace, void* stream);
/* Actor.infer_priv_latent (actor_critic.py:219-221) for every row: params = priv_encoder.{0,2}.{weight,bias} ([64,24], [64],
 * [20,64], [20]); obs f32 [rows, 860]; out f32 [rows, 20]. */
int wbc_priv_latent(const void* const* params, const float* obs, float* out, int rows, void* stream);
int wbc_ppo_grad_floats(void);
int wbc_ppo_num_splits(void);
size_t wbc_ppo_workspace_floats(int B);

/* What Isaac Gym's mass-matrix / Jacobian tensors give the torque-supervision path (widowGo1.py:550-558, 1201-1242):
 * mm f32 [N,6,6] = joint-space inertia block of the 6 arm joints, jac f32 [N,6,6] = world-frame Jacobian of the
 * end-effector rigid body w.r.t. them (rows linear xyz, angular xyz), gtorque f32 [N,6] = sum over the rigid bodies
 * link_rb9 (the actor's last 9) of J_body^T (0,0,9.81 m,0,0,0) with the host masses link_mass9 (+ env 0's gripper mass
 * delta, as the reference reads env 0's properties). Computed from the sim's current root / DoF state. */
int wbc_sim_arm_dynamics(wbc_sim* sim, const int* link_rb9, const float* link_mass9, float* mm, float* jac,
                         float* gtorque, void* stream);

/* extras["episode"] of reset_idx (widowGo1.py:743-754): out[0:WBC_NREW] = mean over the envs that reset in the last
 * step of their finished episode's reward sums, out[WBC_NREW:+WBC_NMETRIC] the same for the metric sums, both
 * times `scale` (1 / max_episode_length_s). `out`: device, WBC_NREW + WBC_NMETRIC floats. On a step in which no env reset
 * the reference does not touch extras["episode"] (reset_idx returns early, widowGo1.py:705-706), i.e. the previous values
 * stay published: out = prev then (prev: the previous call's output, or NULL = zeros). */
int wbc_sim_episode_stats(wbc_sim* sim, float scale, const float* prev, float* out, void* stream);
/* The same launch with one more workgroup that does wbc_runner_track_episodes' work (below) on this sim's reward / reset
 * buffers: the logged training loop's episode bookkeeping at no launch of its own. track_state == NULL: wbc_sim_episode_stats. */
int wbc_sim_episode_stats_track(wbc_sim* sim, float scale, const float* prev, float* out, float* track_state, int track_cap,
                                void* stream);

/* PPO.process_env_step's tensor work (rsl_rl/algorithms/ppo.py:129-141 + rollout_storage.py:70-72) in one launch:
 * out_rewards[n] = (rew[n], arm_rew[n]) + gamma * values[n] * time_outs[n], out_dones[n] = dones[n] != 0.
 * time_outs may be NULL (no bootstrap). All pointers device; out_* are the rollout-storage slots of this step. */
int wbc_rollout_store(const float* rew, const float* arm_rew, const int64_t* dones, const uint8_t* time_outs,
                      const float* values, float gamma, float* out_rewards, uint8_t* out_dones, int n, void* stream);

/* OnPolicyRunner.learn's per-step episode bookkeeping (rsl_rl/runners/on_policy_runner.py:140-154: cur_reward_sum +=
 * rewards, cur_episode_length += 1, new_ids = (dones > 0).nonzero(), rewbuffer / armrewbuffer / lenbuffer.extend(...),
 * donebuffer.append(len(new_ids) / N), the sums of new_ids zeroed) in one launch and without the reference's three host
 * copies per env step. `state` (device f32, wbc_runner_track_state_floats(n, cap), zero-initialised by the caller):
 *   cur[n][3] running (reward, arm reward, length) | ring[cap][3] the last `cap` finished episodes in the order the
 *   reference's deques receive them (step by step, ascending env index) | done_ring[cap] the last `cap` per-step done
 *   fractions | 4 int32: ring head, ring fill (<= cap), done_ring head, done_ring fill.
 * cap = the deques' maxlen (100). The host reads ring / done_ring once per iteration. */
int wbc_runner_track_episodes(const float* rew, const float* arm_rew, const int64_t* dones, int n, int cap, float* state,
                              void* stream);
size_t wbc_runner_track_state_floats(int n, int cap);

/* Per-env shape (up to 3 dims; the leading N is implied), ndim and dtype (0 f32, 1 i64, 2 u8) of tensor `id`, without a sim. */
int wbc_tensor_spec(int id, int64_t* dims3, int* ndim, int* dtype);

/* sizeof(wbc_model), sizeof(wbc_task_cfg), sizeof(wbc_curriculum): lets a binding check its mirrors. */
void wbc_abi_sizes(int* out3);

/* LeggedRobot._get_heights (legged_gym/envs/base/legged_robot.py:793-829): terrain height under num_points points around
 * every robot -- points rotated by the base yaw (utils/math.py:38-42) and offset by the base position, `+= border_size`,
 * `(p / horizontal_scale).long()` (toward zero), clip to [0, dim-2], min of the three corner samples, * vertical_scale.
 * Bit-exact against oracle/terrain_oracle.py (integer / index work; the divisor is applied as PyTorch's CUDA kernel does:
 * multiplication by the fp32 reciprocal). base_quat: device f32 rows of (x,y,z,w), quat_stride floats apart; root_pos:
 * device f32 rows starting with (x,y), pos_stride floats apart (root_states: 13); height_points: device f32 [N,P,3];
 * height_samples: device int16 [rows,cols]; out: device f32 [N,P]. */
int wbc_get_heights(const float* base_quat, int quat_stride, const float* root_pos, int pos_stride, const float* height_points,
                    const int16_t* height_samples, int rows, int cols, float border_size, float horizontal_scale,
                    float vertical_scale, float* out, int num_envs, int num_points, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WBC_SIM_H */
