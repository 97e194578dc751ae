/*
 * dtsim.h -- C-ABI of libdtsim.so, the MI355X-native batched Duckietown step path.
 *
 * The reference (duckietown/gym-duckietown v6.1.34) has no FFI boundary: its hot
 * path is in-process Python (src/gym_duckietown/simulator.py).  This header is the
 * boundary a maintainer would bind *below* the reference's gym.Env surface; every
 * entry point cites the reference code it replaces (paths under /root/reference/).
 * The ctypes binding is gym-duckietown_amd/dtsim/_ffi.py; INTEGRATION.md shows the
 * stub the reference's simulator.py would add.
 *
 * Conventions
 *   - plain C types, caller-owned inputs (the library copies), no torch types;
 *   - every function returns 0 (DTSIM_OK) or a negative error code and never throws;
 *     dtsim_last_error() returns a thread-local message for the last failure;
 *   - one handle per device; a handle is not thread-safe, distinct handles are
 *     independent; all work is stream-ordered on the handle's HIP stream;
 *   - frames / SoA fields live in device memory owned by the library (or bound by
 *     the caller with dtsim_bind_frames) and are exposed as raw device pointers.
 *   - world frame = the reference's "weird" frame (simulator.py:1629-1652):
 *     x right (tile column i), y up, z = tile row j; heading theta gives
 *     dir = (cos t, 0, -sin t) (simulator.py:2056-2073).
 */
#ifndef DTSIM_H
#define DTSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTSIM_ABI_VERSION 11

/* error codes */
#define DTSIM_OK 0
#define DTSIM_E_INVALID (-1)   /* bad argument */
#define DTSIM_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define DTSIM_E_NOGPU (-3)     /* no HIP device: the library has no CPU fallback */
#define DTSIM_E_STATE (-4)     /* call sequence error (e.g. step before set_maps/reset) */
#define DTSIM_E_LIMIT (-5)     /* a compile-time limit below was exceeded */

/* limits (device tables are LDS-staged) */
#define DTSIM_MAX_MAPS 32
#define DTSIM_MAX_TILES 1024        /* grid_w * grid_h per map */
#define DTSIM_MAX_CURVES 1024       /* per map */
#define DTSIM_MAX_STATIC 56         /* collidable static objects per map */
#define DTSIM_MAX_DYNAMIC 8         /* dynamic objects (DuckieObj + DuckiebotObj) per map */
#define DTSIM_MAX_OBJECTS 64        /* renderable objects per map */
#define DTSIM_MAX_DELAY 16          /* dynamics command delay, in steps (0.15 s: up to 106 Hz) */
#define DTSIM_MAX_TEXTURES 96
#define DTSIM_MAX_MESHES 64

/* dtsim_config.flags */
#define DTSIM_F_RENDER 1u        /* allocate the [N,H,W,3] frame buffer */
#define DTSIM_F_DISTORTION 2u    /* fisheye: needs dtsim_set_distortion_lut (distortion.py:85-125) */
#define DTSIM_F_DOMAIN_RAND 4u   /* per-env camera/light/colour parameters are honoured (simulator.py:551-614) */
#define DTSIM_F_AUTO_RESET 8u    /* envs whose done flag is set restart from the spawn pool at the next step */
#define DTSIM_F_ACTIONS_F64 16u  /* dtsim_step actions are double[...] instead of float[...] */
#define DTSIM_F_PROFILE 32u      /* bracket every kernel launch with HIP events (dtsim_profile_read) */
#define DTSIM_F_LIGHT_CAPTURE 64u /* (ABI v11) device-side resets -- DTSIM_F_AUTO_RESET and dtsim_reset(states = NULL) -- position the new episode's light as GL
                                  * does: reset() calls glLightfv(GL_POSITION) with whatever model-view the LAST FRAME left (simulator.py:565-584), so from the
                                  * second episode on the light given in the init state is taken through the camera of the pose the previous episode ended at
                                  * (a direction is rotated, a position also translated) before it becomes the env's eye-space light.  Honoured by the per-env
                                  * render path (DTSIM_F_DOMAIN_RAND); dtsim_reset(states) takes the light as given (the caller's business, as the gym facade
                                  * does it on the host). */

/* dtsim_config.action_mode */
#define DTSIM_ACTION_WHEELS 0    /* Simulator.step: [left, right] duty, clipped to [-1,1] (simulator.py:1669-1672) */
#define DTSIM_ACTION_VEL_STEER 1 /* DuckietownEnv.step: (vel, steering) -> duty (envs/duckietown_env.py:36-61) */

/* done codes (simulator.py:1685-1705 DoneRewardInfo.done_code) */
#define DTSIM_DONE_IN_PROGRESS 0
#define DTSIM_DONE_INVALID_POSE 1
#define DTSIM_DONE_MAX_STEPS 2

/* tile kinds (simulator.py:840-848 + the non-drivable kinds of the map format) */
enum {
  DTSIM_TILE_EMPTY = 0, /* "empty": no tile (simulator.py:820) */
  DTSIM_TILE_STRAIGHT = 1,
  DTSIM_TILE_CURVE_LEFT = 2,
  DTSIM_TILE_CURVE_RIGHT = 3,
  DTSIM_TILE_3WAY_LEFT = 4,
  DTSIM_TILE_3WAY_RIGHT = 5,
  DTSIM_TILE_4WAY = 6,
  DTSIM_TILE_ASPHALT = 7,
  DTSIM_TILE_GRASS = 8,
  DTSIM_TILE_FLOOR = 9,
  DTSIM_TILE_OTHER = 10
};

typedef struct dtsim dtsim_t;

/* Constructor arguments of Simulator / DuckietownEnv that reach the hot path
 * (simulator.py:207-232, envs/duckietown_env.py:15-34). */
typedef struct dtsim_config {
  uint32_t struct_size;  /* sizeof(dtsim_config), ABI check */
  uint32_t flags;        /* DTSIM_F_* */
  int32_t num_envs;
  int32_t device;        /* HIP device ordinal */
  int32_t cam_width;     /* camera_width  (default 640) */
  int32_t cam_height;    /* camera_height (default 480) */
  int32_t frame_skip;    /* simulator.py:1673 */
  int32_t max_steps;     /* simulator.py:1694 */
  int32_t delay_steps;   /* duckietown_world ApplyDelay(0.15 s) in steps; 5 at 30 Hz (PARITY UNPINNED) */
  int32_t action_mode;   /* DTSIM_ACTION_* */
  double delta_time;     /* 1 / frame_rate (simulator.py:300) */
  double robot_speed;    /* constant used by the reward (simulator.py:1702) */
  double gain, trim, radius, k, limit; /* DuckietownEnv kinematics (envs/duckietown_env.py:15) */
  void* stream;          /* hipStream_t to launch on; NULL = library creates one */
} dtsim_config;

/* One world object as interpreted by Simulator.interpret_object (simulator.py:933-1038)
 * and WorldObj.__init__ / DuckieObj.__init__ (objects.py:33-66, 339-365).  OBB corners
 * and SAT axes are computed on the host exactly as the reference does
 * (generate_corners collision.py:64-79, generate_norm collision.py:99-106). */
typedef struct dtsim_object {
  int32_t mesh_id;       /* index into dtsim_set_assets meshes, -1 = not rendered */
  int32_t dynamic;       /* 0 static WorldObj, 1 DuckieObj pedestrian (objects.py:339), 2 DuckiebotObj follower (objects.py:180),
                            3 CheckerboardObj (objects.py:479-587: scripted calibration motion, `vel` carries the initial step counter) */
  int32_t collidable;    /* static && kind != trafficlight (simulator.py:1027-1030) */
  int32_t optional;
  double pos[3];
  double angle;          /* radians */
  double scale;
  double corners[8];     /* [4][2] (x,z) */
  double norm[4];        /* [2][2] */
  double safety_radius;
  double spawn_clear;    /* max(max_coords)*0.5*scale + MIN_SPAWN_OBJ_DIST (simulator.py:1467) */
  /* DuckieObj parameters (objects.py:339-365); ignored for static objects.  For a DuckiebotObj the same
   * four slots carry follow_dist, velocity, gain, trim (objects.py:199-216); its radius / k / limit /
   * wheel_dist / robot_width / robot_length are the reference defaults. */
  double walk_distance, vel, wait_time, wiggle;
  /* TrafficLightObj (objects.py:434-477): every `light_freq` seconds of object time the first material
   * chunk of the mesh (its first `light_tris` triangles) switches between textures light_tex[0] and
   * light_tex[1]; light_pattern is the initial pattern.  light_freq = 0: not a traffic light. */
  int32_t light_freq, light_pattern;
  int32_t light_tex[2];
  int32_t light_tris, light_pad;
} dtsim_object;

typedef struct dtsim_map {
  int32_t grid_w, grid_h;          /* simulator.py:801-802 */
  double tile_size;                /* road_tile_size */
  const uint8_t* tile_kind;        /* [grid_h*grid_w] DTSIM_TILE_*, index j*grid_w+i */
  const uint8_t* tile_angle;       /* [..] 0..3, index into [S,E,N,W] (simulator.py:823-830) */
  const int16_t* tile_tex;         /* [..] texture id (dtsim_set_assets) or -1 */
  const int16_t* tile_curve_off;   /* [..] first curve of the tile, -1 if not drivable */
  const uint8_t* tile_curve_cnt;   /* [..] 2 / 6 / 12 (simulator.py:1151-1335) */
  int32_t n_curves;
  const double* curves;            /* [n_curves][4][2] Bezier control points (x,z) */
  const double* curve_heads;       /* [n_curves][2] chord P3-P0 divided by the tile's Frobenius norm,
                                      computed on the host with the reference's numpy expression
                                      (simulator.py:1355-1356) */
  int32_t n_objects;
  const dtsim_object* objects;     /* [n_objects] in map order */
} dtsim_map;

/* RGBA8 texture, row 0 = v=0 (bottom row of the image, the GL/pyglet origin). */
typedef struct dtsim_texture {
  int32_t width, height;           /* powers of two */
  const uint8_t* rgba;             /* [height][width][4] */
} dtsim_texture;

/* Triangle soup of one mesh after ObjMesh's recentring (objmesh.py:181-232). */
typedef struct dtsim_mesh {
  int32_t n_tris;
  const float* verts;              /* [n_tris][3][3] */
  const float* normals;            /* [n_tris][3][3] */
  const float* colors;             /* [n_tris][3][3] per-vertex Kd */
  const float* uvs;                /* [n_tris][3][2] texture coordinates (objmesh.py:199-207), or NULL */
  const int32_t* tri_tex;          /* [n_tris] texture index of the triangle's material chunk (map_Kd,
                                      objmesh.py:268-275: GL_LINEAR / GL_REPEAT, MODULATE with the lit
                                      vertex colour), -1 = untextured; NULL = all untextured */
} dtsim_mesh;

/* Everything Simulator.reset() decides for one env (simulator.py:528-763).  Drawn on
 * the host in the reference's RNG order (dtsim/reset.py) -- or by the oracle in parity
 * tests -- and uploaded. */
#define DTSIM_MAP_RELOAD 0x40000000

typedef struct dtsim_init_state {
  double pos[3];            /* cur_pos  simulator.py:740 */
  double angle;             /* cur_angle simulator.py:741 */
  int32_t map_id;           /* index into dtsim_set_maps; | DTSIM_MAP_RELOAD: re-create the map's objects even if the env
                               is already on this map (randomize_maps_on_reset reloads the map at every reset,
                               simulator.py:541-544) */
  int32_t dynamics_trim_on; /* dynamics_rand: get_DB18_uncalibrated(trim) simulator.py:746-748 */
  double dynamics_trim;
  double wheel_dist;        /* simulator.py:597 */
  double cam_height;        /* simulator.py:602,612 */
  double cam_angle_deg;     /* simulator.py:605,613 */
  double cam_fov_y_deg;     /* simulator.py:608,614 */
  double camera_noise[3];   /* simulator.py:1768-1769 (applied only with DTSIM_F_DOMAIN_RAND) */
  double horizon_color[3];  /* simulator.py:551-562 */
  double ground_color[3];   /* simulator.py:594 */
  double light_pos[4];      /* simulator.py:565-570, w=0 directional / w=1 positional */
  double light_ambient[3];  /* simulator.py:573-574 */
  double light_diffuse[3];  /* simulator.py:575-576 */
} dtsim_init_state;

/* Result of evaluating the reference's geometry queries at an arbitrary pose. */
typedef struct dtsim_probe {
  int32_t tile_i, tile_j;   /* get_grid_coords simulator.py:1134 */
  int32_t curve_idx;        /* argmax curve of closest_curve_point simulator.py:1362, -1 */
  uint8_t drivable;         /* _drivable_pos(pos) simulator.py:1411 */
  uint8_t collision;        /* _collision(get_agent_corners(pos, angle)) simulator.py:1473 */
  uint8_t valid;            /* _valid_pose(pos, angle, safety_factor) simulator.py:1494 */
  uint8_t in_lane;          /* get_lane_pos2 did not raise NotInLane */
  uint8_t inconvenient;     /* _inconvenient_spawn(pos) simulator.py:1461 */
  uint8_t pad[3];
  double t;                 /* bezier_closest graphics.py:316 */
  double point[2], tangent[2]; /* closest_curve_point (x,z) */
  double dist, dot_dir, angle_deg, angle_rad; /* LanePosition simulator.py:1371-1409 */
  double prox;              /* proximity_penalty2 simulator.py:1430 */
  double reward;            /* compute_reward(pos, angle, robot_speed) simulator.py:1654 */
} dtsim_probe;

/* SoA fields for dtsim_read / dtsim_write / dtsim_field_devptr. */
enum {
  DTSIM_FIELD_POS = 0,        /* double [N][3]  cur_pos */
  DTSIM_FIELD_ANGLE = 1,      /* double [N]     cur_angle */
  DTSIM_FIELD_REWARD = 2,     /* double [N] */
  DTSIM_FIELD_DONE = 3,       /* uint8  [N] */
  DTSIM_FIELD_DONE_CODE = 4,  /* uint8  [N] */
  DTSIM_FIELD_STEP_COUNT = 5, /* int32  [N] */
  DTSIM_FIELD_TILE = 6,       /* int32  [N][2] get_grid_coords(cur_pos) */
  DTSIM_FIELD_LANE = 7,       /* double [N][4] dist, dot_dir, angle_deg, angle_rad (0 if not in lane) */
  DTSIM_FIELD_IN_LANE = 8,    /* uint8  [N] */
  DTSIM_FIELD_PROX = 9,       /* double [N] proximity_penalty2 */
  DTSIM_FIELD_SPEED = 10,     /* double [N] simulator.py:1568 */
  DTSIM_FIELD_TIMESTAMP = 11, /* double [N] */
  DTSIM_FIELD_WHEELS = 12,    /* double [N][2] last [left,right] duty passed to the dynamics */
  DTSIM_FIELD_MAP_ID = 13,    /* int32  [N] */
  DTSIM_FIELD_OBJ_CENTER = 14,/* double [N][DTSIM_MAX_DYNAMIC][2] DuckieObj.center (x,z) */
  DTSIM_FIELD_OBJ_ACTIVE = 15,/* uint8  [N][DTSIM_MAX_DYNAMIC] pedestrian_active */
  DTSIM_FIELD_OBJ_YROT = 16,  /* double [N][DTSIM_MAX_DYNAMIC] y_rot in degrees */
  DTSIM_FIELD_OBJ_PARAMS = 17,/* double [N][DTSIM_MAX_DYNAMIC][3] vel, wait_time, wiggle (write = DR override) */
  DTSIM_FIELD_OBJ_VISIBLE = 18,/* uint8 [N][DTSIM_MAX_OBJECTS] obj.visible (simulator.py:653-656) */
  DTSIM_FIELD_EPISODE = 19,   /* int32  [N] episodes started (auto-reset counter) */
  DTSIM_FIELD_STATE_BLOB = 20,/* opaque: full SoA state, dtsim_state_bytes() bytes (checkpoint) */
  DTSIM_FIELD_OBJ_LIGHT = 21, /* uint8  [N][DTSIM_MAX_OBJECTS] TrafficLightObj.pattern (0 for other objects) */
  DTSIM_FIELD_OBJ_Y = 22,     /* double [N][DTSIM_MAX_DYNAMIC] height of the object centre (CheckerboardObj moves in y) */
  DTSIM_FIELD_OBJ_EXTRA = 23, /* double [N][DTSIM_MAX_DYNAMIC][5] DuckiebotObj follow_dist, radius, wheel_dist, robot_width,
                               * robot_length (objects.py:198-215; its velocity, gain, trim are OBJ_PARAMS) */
  DTSIM_FIELD_CAMERA = 24,    /* float  [N][6]  cam_height, cam_angle[0] (rad), cam_fov_y (rad), camera_noise xyz (simulator.py:596-614) */
  DTSIM_FIELD_COLORS = 25,    /* float  [N][16] horizon rgb, ground rgb, light ambient rgb, diffuse rgb, light_pos xyzw (:551-594) */
  DTSIM_FIELD_WHEEL_DIST = 26,/* double [N]     wheel_dist (:597) */
  DTSIM_FIELD_RENDER_POS = 27,/* int32  [N]     read-only: position of each env in the render order of the last dtsim_render (k_env_sort:
                               * envs standing on the same tile and facing the same way are neighbours; 32 consecutive positions share a
                               * raster workgroup, XCD x owns the x-th eighth of the order); the identity when the pass ran in index order */
  DTSIM_FIELD__COUNT = 28
};

/* kernels for dtsim_profile_read */
enum { DTSIM_KERNEL_STEP = 0, DTSIM_KERNEL_RENDER = 1, DTSIM_KERNEL_RESET = 2, DTSIM_KERNEL_QUERY = 3, DTSIM_KERNEL_OBSERVE = 4, DTSIM_KERNEL__COUNT = 5 };

int dtsim_abi_version(void);
const char* dtsim_last_error(void);
/* Number of visible HIP devices, or a negative error code. */
int dtsim_device_count(void);

/* Simulator.__init__ (simulator.py:207-384), minus GL. */
int dtsim_create(const dtsim_config* cfg, dtsim_t** out);
void dtsim_destroy(dtsim_t* h);

/* Textures (graphics.py:69-169 load_texture) and meshes (objmesh.py:62-293). */
int dtsim_set_assets(dtsim_t* h, const dtsim_texture* textures, int n_textures,
                     const dtsim_mesh* meshes, int n_meshes);
/* Simulator._interpret_map output (simulator.py:788-1038), host-prepared. */
int dtsim_set_maps(dtsim_t* h, const dtsim_map* maps, int n_maps);
/* Distortion.rmapx/rmapy (distortion.py:100-111): [cam_height][cam_width] float each.
 * The per-frame cv2.remap(INTER_NEAREST) (distortion.py:118) is folded into the raster. */
int dtsim_set_distortion_lut(dtsim_t* h, const float* rmapx, const float* rmapy);

/* Simulator.reset() for the envs with mask[e] != 0 (mask NULL = all): state taken from
 * states[e] (array of num_envs entries). */
int dtsim_reset(dtsim_t* h, const uint8_t* mask, const dtsim_init_state* states);   /* states == NULL: device sampler */
/* Device-side reset sampler.  The reference's reset() draws, per env, the domain-
 * randomisation values (simulator.py:546-614, randomizer.py:36-91), the visibility of optional objects
 * (:648-656), a start tile (:659-676) and then poses until one is accepted (:692-738: not an
 * inconvenient spawn, valid at safety factor 1.3, within accept_start_angle_deg of the lane).  With a
 * sampler installed the same *distributions* and acceptance test run on the device from a counter-based
 * generator (Philox4x32-10 keyed by seed and env index, counter = episode) -- the stream differs from
 * numpy's PCG64, so this mode is for throughput, not for RNG-order parity (dtsim_reset with host states
 * / the spawn pool keep that).  Used by DTSIM_F_AUTO_RESET when installed (it takes precedence over the
 * pool) and by dtsim_reset(h, mask, NULL). */
typedef struct dtsim_reset_sampler {
  uint64_t seed;
  int32_t domain_rand;              /* draw camera / light / colour / wheel_dist perturbations */
  int32_t dynamics_rand;            /* apply the drawn trim (simulator.py:744-750) */
  int32_t map_cycle;                /* 1: MultiMapEnv, the next map at every reset (multimap_env.py:44-49);
                                       2: randomize_maps_on_reset, a uniformly drawn map, reloaded (simulator.py:541-544) */
  int32_t max_attempts;             /* MAX_SPAWN_ATTEMPTS, 5000 */
  double accept_start_angle_deg;    /* simulator.py:724-728 */
  double color_sky[3], color_ground[3];
  int32_t start_tile[DTSIM_MAX_MAPS][2];  /* user_tile_start / the map's start_tile; -1 = a random drivable tile */
  int32_t has_start_pose[DTSIM_MAX_MAPS]; /* the map defines start_pose: spawn at tile * tile_size + pose, no rejection loop */
  double start_pose[DTSIM_MAX_MAPS][3];   /* (x offset, z offset, angle): simulator.py:679-688 */
} dtsim_reset_sampler;
int dtsim_set_reset_sampler(dtsim_t* h, const dtsim_reset_sampler* sampler);   /* NULL uninstalls */
/* Reset, with the installed sampler, every env whose done flag is set (the mask is read on the device: no
 * host round trip).  What a vectorised learner loop does right after a step: read reward / done, then
 * restart the finished episodes so that the observation it gets is the new episode's first one. */
int dtsim_reset_done(dtsim_t* h);

/* Pool of spawn states used by DTSIM_F_AUTO_RESET: env e starts episode k from
 * pool[(e + k * num_envs) % n_pool]. */
int dtsim_set_spawn_pool(dtsim_t* h, const dtsim_init_state* pool, int n_pool);

/* n_steps x { [DuckietownEnv.step ->] Simulator.step } without the render
 * (simulator.py:1669-1705; update_physics :1551; _update_pos :2076; _compute_done_reward :1685).
 * actions: [n_steps][num_envs][2], float (or double with DTSIM_F_ACTIONS_F64), host or
 * device pointer (actions_on_device).  Asynchronous. */
int dtsim_step(dtsim_t* h, const void* actions, int n_steps, int actions_on_device);

/* dtsim_step with flags; dtsim_step(...) == dtsim_step_ex(..., 0).
 *   DTSIM_STEP_ONE_UPDATE  one `Simulator.update_physics(action)` (simulator.py:1551-1584) per step: like dtsim_step but
 *                          the action is the wheel pair update_physics takes -- no (vel, steering) kinematics
 *                          (DuckietownEnv.step's) and no np.clip (Simulator.step's, :1670) -- and
 *                          frame_skip is not applied (step_count += 1, timestamp += delta_time, speed, every object
 *                          stepped once; reward / done are refreshed for the new state -- they are pure functions of it,
 *                          `_compute_done_reward` :1685 evaluates them on demand in the reference).
 *   DTSIM_STEP_POSE_ONLY   the module-level `_update_pos(self, action)` (simulator.py:2076-2088): the dynamics state
 *                          (incl. the delayed-command queue) advances by one delta_time and the resulting pose is
 *                          written to the pose fields; step_count, timestamp, speed, objects, reward and done are
 *                          left alone, DuckietownEnv's (vel, steering) kinematics and auto-reset do not apply (the
 *                          action is the wheel pair `_update_pos` takes, simulator.py:2083). */
enum { DTSIM_STEP_ONE_UPDATE = 1u, DTSIM_STEP_POSE_ONLY = 2u };
int dtsim_step_ex(dtsim_t* h, const void* actions, int n_steps, int actions_on_device, uint32_t flags);

/* Simulator.render_obs (simulator.py:1953-1972) = _render_img (:1707-1951) + distort:
 * writes [num_envs][cam_height][cam_width][3] uint8, row 0 = image top.  Asynchronous. */
int dtsim_render(dtsim_t* h);

/* Segmentation render, `_render_img(..., segment=True)` (simulator.py:1730-1737, 1753, 1808, 1879; reached from
 * reset(segment) :760, render_obs(segment) :1953 and render(mode, segment) :1974): lighting off, colour buffer
 * cleared to and ground quad drawn in magenta, every texture replaced by its segmented version
 * (graphics.py:52-57 Texture.bind -> load_texture(segment=True) :70-126), every mesh drawn through
 * get_mesh(name, segment=True) = flat `gen_segmentation_color(mesh_name)` modulated by the per-vertex Kd
 * (objmesh.py:255-292, 360-362).  The segmented textures are prepared on the host (dtsim/assets.py) and must
 * mirror the dtsim_set_assets list entry for entry (same count, same sizes); mesh_rgb is [n_meshes][3]. */
int dtsim_set_segment_assets(dtsim_t* h, const dtsim_texture* textures, int n_textures,
                             const uint8_t* mesh_rgb, int n_meshes);
enum { DTSIM_RENDER_SEGMENT = 1u, DTSIM_RENDER_GL_FILTER = 2u };
/* dtsim_render with flags (DTSIM_RENDER_*); dtsim_render(h) == dtsim_render_ex(h, 0).
 * DTSIM_RENDER_GL_FILTER (ABI v11): this pass filters tile textures with the arithmetic of the reference's renderer -- Mesa llvmpipe's 8-bit
 * GL_LINEAR: two lerps with an 8-bit intermediate (csrc/render.hip gl_linear_rgb) -- instead of the quad-record kernels' one-pass byte-weight
 * filter: the generic raster runs, 2 - 4 x slower, and the frames are bit-identical to the reference's on 99.2 - 99.96 % of the pixels instead of
 * within +-1/255 on all but 0.25 % (tests/test_gpu_gl_golden.py).  For validation against reference frames, not for throughput. */
int dtsim_render_ex(dtsim_t* h, uint32_t flags);
/* GL_LINE overlays of the reference -- draw_curve (simulator.py:1886-1904, graphics.py:336-349) and draw_bbox (simulator.py:1907-1918,
 * objects.py:131-146) -- as a post-pass on the frames the last dtsim_render made: `lines` = [n][9] floats, world-space segment
 * (ax, ay, az, bx, by, bz) + colour (r, g, b in 0..1, the glColor3f of the line); env_idx[i] = the env segment i is drawn into (NULL: all
 * into env 0; must be non-decreasing).  Each segment goes through the env's camera, is clipped at the near plane and rasterised as a
 * 1-pixel line under 4x multisampling (coverage per sample, first line wins a sample); the colour is the glColor lit as a surface with
 * normal +y.  Lines are not depth-tested against mesh objects (they lie above the tile plane).  Synchronous (host segments).
 * Both post-passes (this and dtsim_draw_leds) use the cameras / projected triangles of THE LAST dtsim_render: dtsim_step, dtsim_reset,
 * dtsim_reset_done and dtsim_write invalidate them -- the call then fails with DTSIM_E_STATE until the state has been rendered again. */
int dtsim_draw_lines(dtsim_t* h, const float* lines, const int32_t* env_idx, int n);
/* The LED spheres of enable_leds (objects.py:68-121: per LED of a duckiebot-kind object a 1 cm gluSphere at alpha 1 and a halo at alpha 0.2,
 * additively blended -- glBlendFunc(GL_SRC_ALPHA, GL_ONE) -- with depth test and depth writes, lit, untextured) as a post-pass on the frames
 * the last dtsim_render made: `spheres` = [n][8] floats, world-space centre (x, y, z), radius, glColor (r, g, b in 0..1), alpha, in DRAW
 * ORDER; env_idx as for dtsim_draw_lines.  Per MSAA sample the depth of the opaque scene is recomputed (planes analytically, meshes from
 * the projected triangles the render pass left) and every sphere whose front surface is nearer adds alpha x its lit colour and writes its
 * depth; the pixel becomes frame + sum / 4.  Deviations from the GL result (it depends on gluSphere's strip order): analytic spheres, front
 * surfaces only, all spheres after all opaque objects.  Needs a preceding dtsim_render of the same state (not the segment view).
 * Synchronous (host spheres). */
int dtsim_draw_leds(dtsim_t* h, const float* spheres, const int32_t* env_idx, int n);
void* dtsim_frames_devptr(dtsim_t* h);
size_t dtsim_frames_bytes(const dtsim_t* h);
/* Render into caller-owned device memory instead (e.g. a torch tensor that is the
 * send buffer of the RCCL all-gather). NULL restores the internal buffer. */
int dtsim_bind_frames(dtsim_t* h, void* devptr);
/* The north star's exchange (no counterpart in the reference, which runs one env per process):
 * RCCL all-gather of this rank's uint8 frame batch (the buffer dtsim_frames_devptr / dtsim_bind_frames names) into
 * `recv` = device memory for n_ranks * dtsim_frames_bytes(h) bytes, rank-major, enqueued on the handle's stream behind
 * the last render (stream-ordered: dtsim_sync or the next call on the handle waits for it).  `nccl_comm` is an
 * ncclComm_t the CALLER created (one process per GPU; ncclCommInitRank in C, or the communicator of its framework);
 * librccl.so is resolved with dlopen at the first call -- the library has no link-time dependency on it, and the
 * call fails with DTSIM_E_STATE where it is absent.  `send` = NULL gathers the frame batch; otherwise `send` /
 * `send_bytes` name another device buffer of this rank to gather instead (e.g. the dtsim_observe output: 57.6 KB per
 * env at 160 x 120 instead of 921.6 KB).  ORDERING: the collective is ordered behind earlier work on the HANDLE's stream only.
 * `recv` (and a caller-owned `send`) must be ready on that stream when the call is made -- a buffer another stream is still
 * filling or zeroing must be synchronised first (host wait, or an event the handle's stream waits on: dtsim_stream()); readers on
 * another stream wait for the handle's stream likewise.  dtsim/sharding.py: ShardedSimulator.gather_frames() uses this entry point
 * on RCCL jobs (events both ways, no host wait). */
int dtsim_allgather_frames(dtsim_t* h, void* nccl_comm, void* recv, const void* send, size_t send_bytes);

/* Learner-side observation of the rendered frame batch, on the device (what the reference's learners do
 * per env on the host): ResizeWrapper (learning/utils/wrappers.py:38-54, scipy imresize == PIL
 * Image.resize BILINEAR, reproduced bit-exactly: Pillow's two-pass 22-bit fixed-point resampler with its
 * uint8 intermediate), optionally ImgWrapper (HWC -> CHW, :72-86) and NormalizeWrapper (/ 255 -> float32,
 * :57-69).  `taps_*` are Pillow's per-output-coordinate tables (first tap, tap count / fixed-point taps)
 * as built by dtsim/resample.py; pass NULL tables for an axis whose size does not change.
 * out: device pointer to [num_envs][out_h][out_w][3] (DTSIM_OBS_HWC) or [num_envs][3][out_h][out_w]
 * (DTSIM_OBS_CHW), uint8 or float32 (DTSIM_OBS_F32).  Asynchronous, stream-ordered after dtsim_render. */
enum { DTSIM_OBS_HWC = 0, DTSIM_OBS_CHW = 1, DTSIM_OBS_F32 = 2 };
int dtsim_observe(dtsim_t* h, void* out, int out_h, int out_w, int flags,
                  const int32_t* bounds_x, const int32_t* taps_x, int ksize_x,
                  const int32_t* bounds_y, const int32_t* taps_y, int ksize_y);

/* The reference's own ResizeWrapper (src/gym_duckietown/wrappers.py:129-138): cv2.resize(obs, (out_w, out_h),
 * interpolation=cv2.INTER_CUBIC) of every frame of the batch, on the device -- OpenCV's 8-bit fixed-point path as
 * published in imgproc/resize.cpp: four taps per axis (A = -0.75) in 11-bit fixed point, replicated borders, int32
 * rows without intermediate rounding, saturate_cast<uchar>((v + 2^21) >> 22).  first_*[i] = source index of the first
 * of the four taps of output coordinate i (may be negative), taps_*: [out][4], both as built by dtsim/resample.py
 * cubic_coeffs (host pointers).  out / flags as dtsim_observe.  Asynchronous, stream-ordered after dtsim_render. */
int dtsim_observe_cubic(dtsim_t* h, void* out, int out_h, int out_w, int flags,
                        const int32_t* first_x, const int32_t* taps_x, const int32_t* first_y, const int32_t* taps_y);

/* Geometry queries of the reference at arbitrary poses, evaluated on the device against
 * env env_idx[q]'s world (its map, dynamic objects and visibility): _valid_pose,
 * _collision, _drivable_pos, get_lane_pos2, closest_curve_point, proximity_penalty2,
 * compute_reward, _inconvenient_spawn.  poses: [n][3] = x, z, angle.  Synchronous. */
int dtsim_query(dtsim_t* h, int n, const int32_t* env_idx, const double* poses,
                double safety_factor, dtsim_probe* out);

/* Everything Simulator.step hands back about ONE env besides the observation -- cur_pos / cur_angle / speed /
 * timestamp (simulator.py:1551-1584), _compute_done_reward (:1685-1705), get_agent_info (:1586-1612) -- gathered
 * on the device and copied with one transfer (a 1-env gym loop otherwise pays one dtsim_read per value).
 * Synchronous. */
typedef struct dtsim_agent_info {
  double pos[3], angle, speed, timestamp;
  double wheels[2];         /* last [left, right] duty passed to the dynamics */
  double lane[4];           /* dist, dot_dir, angle_deg, angle_rad (0 if not in lane) */
  double prox, reward;
  int32_t tile[2], step_count;
  uint8_t in_lane, done, done_code, pad;
} dtsim_agent_info;
int dtsim_read_agent(dtsim_t* h, int env, dtsim_agent_info* out);

int dtsim_read(dtsim_t* h, int field, void* dst, size_t bytes);   /* synchronous D2H */
int dtsim_write(dtsim_t* h, int field, const void* src, size_t bytes);
void* dtsim_field_devptr(dtsim_t* h, int field);
size_t dtsim_field_bytes(const dtsim_t* h, int field);
size_t dtsim_state_bytes(const dtsim_t* h);

int dtsim_sync(dtsim_t* h);
void* dtsim_stream(dtsim_t* h);   /* the hipStream_t launches go to */
/* Sum of HIP-event durations of the launches of `kernel` since the last call
 * (requires DTSIM_F_PROFILE).  Synchronises the stream. */
int dtsim_profile_read(dtsim_t* h, int kernel, int* n_launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* DTSIM_H */
